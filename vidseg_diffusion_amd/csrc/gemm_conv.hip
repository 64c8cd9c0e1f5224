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
#include <hip/hip_ext.h>
#include <thread>
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

struct GemmParams {
    // A operand
    const bf16_t* x0;
    const bf16_t* x1;        // second concat source or nullptr
    int C0, C1;              // channels of each source (Cin = C0 + C1)
    long long x0_bytes, x1_bytes;   // extents of the two sources (buffer descriptors: reads beyond them return 0)
    int Hin, Win;            // stored input height/width (conv) ; unused for ksize == 1
    int Hout, Wout;
    int ksize, stride, up;   // ksize 1 or 3; stride 1/2; up 1/2 (nearest upsample of the input before the conv)
    int pad;                 // top/left zero padding of a 3x3 conv: 1, or 0 for the VAE's (0,1,0,1)-padded stride-2 Downsample
                             // (sgm/modules/diffusionmodules/model.py:84-91); bottom/right padding is whatever falls off the input
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
    int res_f32;             // 1: `residual` points to fp32 values (the exact mode's fp32 residual stream, *_rf32 entry points)
    bf16_t* out;             // [M][ldo] or nullptr
    int ldo;
    float* out_f32;          // [M][ldo] or nullptr
    f16* out_split3;         // exact mode, plain or GEGLU linears: the result as the consumer's split operand image [hi | lo | hi]
                             // [M][3 * ldo] = [hi | lo | third plane: unwritten for ldo % 64 == 0], hi = fp16(x), lo = fp16(x - hi); or nullptr
    f16* plane_hi;           // exact mode, the fused q | k | v projection: columns >= plane_col0 leave as the attention kernel's K / V
    f16* plane_lo;           // operand planes hi = fp16(x), lo = fp16(x - hi), [M][plane_ld] each, instead of going to out / out_f32
    int plane_col0, plane_ld;
    f16* tap;                // fp16 copy of columns [0, tap_cols) with leading dim tap_ld, or nullptr
    f16* tap2;               // fp16 copy of columns [tap_cols, 2*tap_cols) (same leading dim), or nullptr
    int tap_cols, tap_ld;
    int tap_early;           // 1: the tap is taken after the bias, BEFORE the per-sample vector / activation (ResBlock.in_layers_features,
                             // openaimodel.py:349-350: in_layers(x) before `h + emb_out`); 0: after them, before the residual
    int act;                 // 0 none, 1 SiLU, 2 GEGLU (32-column interleaved x|gate groups)
    int ksplit;              // >1: K range split over `ksplit` blocks per tile, fp32 partials in ws[split][M][N]
    float* ws;
    int tmode, T;            // tmode: ksize 3 taps run over TIME (Conv3d kernel [3,1,1], video_model.py:45-58): row m is
                             // frame (m / (Hout*Wout)) % T, tap dt reads row m + (dt-1)*Hout*Wout if that frame exists
    int tap_T, tap_S;        // >0: taps are written in the reference's temporal layout [(b s), t, c] (row permutation)
    const float* rowadd;     // per-row scalar added to every column before the residual (attention-output modulation
                             // lambda*mask[:,None], attention.py:646-663, 697-719) or nullptr
    const float* blend;      // exact VideoUNet: after the residual, v = blend_a * blend[m][n] + blend_b * v (AlphaBlender, diffusionmodules/util.py:343-380:
    float blend_a, blend_b;  // alpha * x_spatial + (1 - alpha) * x_temporal with x_temporal = this GEMM's result); fp32 [M][ldo] or nullptr
    int geglu16;             // GEGLU weight rows interleaved in 16-row value | gate groups (k_gemm_p7x<4, true>) instead of 32-row ones
    int split2;              // exact mode: the operands are split images -- A rows [a_hi | a_lo | (unused)] with row stride 3 Cin, W rows
                             // [w_hi | w_hi | w_lo] in the usual K order over 3 * Cin channels (exact.py); k_gemm_p7x / k_gemm_phx stage each plane
                             // once, the other kernels walk the 3 Cin axis and fold its last third back onto plane 0 (a_fold)
    int a_fold;              // split images carry TWO planes: channels [a_fold, 3 Cin / 3 * 3) of the K walk read plane 0 again (a_hi . w_lo);
                             // = 2 * Cin for split2 launches (set by launch_gemm), 0 otherwise.  The third plane of an image is never read.
    int gn;                  // tile columns per panel of the launch order (map_tile)
    int taps, kchunk;        // K order of a conv weight row: k = (c / kchunk) * taps * kchunk + tap * kchunk + c % kchunk.
                             // kchunk = 64 (channel-chunk major: the taps of one 64-channel chunk are consecutive K-tiles, so the
                             // 9 shifted reads of an input pixel's 128-byte line are one K-tile apart and hit L1/L2 instead of
                             // going back to memory 9 times) when Cin and C0 are multiples of 64, else kchunk = Cin (tap major)
};

// Tile order inside an XCD's contiguous chunk of logical ids (map_tile below): panels of gn tile columns, row-major inside a
// panel, so the tiles resident together on an XCD form a (resident/gn) x gn block and share A slabs along rows and W slabs along
// columns through that XCD's L2 (gn = tiles_n is plain row-major; the host picks gn to minimise rows*A_slab + gn*W_slab, see
// launch_gemm).
// blockIdx -> (K split, tile row, tile column) for the LDS-DMA kernels, all in unsigned 32-bit (a launch has < 2^31 blocks; the
// 64-bit divisions this replaces were ~100 scalar instructions each, paid by every block).  Blocks go to XCDs round-robin by
// blockIdx: each XCD gets one contiguous chunk of the (split, tile) space, so a K split's tiles (which share that split's slice of
// W and of A's channels) meet in one L2; inside the chunk the tiles follow panel order.
__device__ __forceinline__ void map_tile(const GemmParams& p, int BM, int BN, int& split, long long& tm, int& tn) {
    const unsigned tiles_m = (unsigned)((p.M + BM - 1) / BM), tiles_n = (unsigned)((p.N + BN - 1) / BN);
    const unsigned nwg = tiles_m * tiles_n;
    unsigned bid = blockIdx.x;
    {
        const unsigned tot = nwg * (unsigned)max(p.ksplit, 1);
        const unsigned q = tot >> 3, r = tot & 7u, xcd = bid & 7u, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const unsigned sp = bid / nwg;
    bid -= sp * nwg;
    split = (int)sp;
    const unsigned gn = (unsigned)p.gn, per = tiles_m * gn, npan = (tiles_n + gn - 1) / gn;
    const unsigned panel = min(npan - 1, bid / per);
    const unsigned pn0 = panel * gn, w = min(gn, tiles_n - pn0), within = bid - panel * per;
    tm = (long long)(within / w);
    tn = (int)(pn0 + within % w);
}

// position of the next K-tile in that order: chunk base channel cq, tap, offset inside the chunk
struct KCursor {
    int cq, tap, sub;
    __device__ __forceinline__ void init(int k, int taps, int kchunk) {
        const int per = taps * kchunk;
        cq = (k / per) * kchunk;
        const int rem = k % per;
        tap = rem / kchunk;
        sub = rem % kchunk;
    }
    __device__ __forceinline__ int c0() const { return cq + sub; }
    __device__ __forceinline__ void advance(int bk, int taps, int kchunk) {
        sub += bk;
        if (sub >= kchunk) {
            sub = 0;
            if (++tap == taps) {
                tap = 0;
                cq += kchunk;
            }
        }
    }
};

#define BK 64

// channel offset of a K-tile inside a split image's pixel: the last third of the 3 Cin walk (against w_lo) reads plane 0 (a_hi) again
__device__ __forceinline__ int fold_c(const GemmParams& p, int c) { return (p.a_fold && c >= p.a_fold) ? c - p.a_fold : c; }

__device__ __forceinline__ long long tap_row(const GemmParams& p, long long m) {
    if (p.tap_T <= 0) return m;
    const long long TS = (long long)p.tap_T * p.tap_S;
    const long long b = m / TS, r = m % TS;
    return (b * p.tap_S + (r % p.tap_S)) * p.tap_T + r / p.tap_S;       // (b t) s -> (b s) t
}
// GELU (erf form, sgm/modules/attention.py:89-115 GEGLU -> F.gelu): erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, three orders
// below the fp16 rounding of the result) with one exp and one reciprocal instead of libm's branching erff: -4..6 % on the GEGLU GEMMs
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float e = 1.0f - poly * __expf(-z * z);
    return 0.5f * x * (1.0f + copysignf(e, x));
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

// one 8-column cell of fp32 results -> the (hi, lo) cells of a split operand image: hi = fp16(x), lo = fp16(x - hi) (common.h)
__device__ __forceinline__ void split_cell8(const float (&v)[8], f16x8& h8, f16x8& l8) {
#if VIDSEG_ACT_IS_F16
    split_hl8(v, h8, l8);
#else
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float x = v[e];
        asm volatile("" : "+v"(x));
        const f16 hh = (f16)x;
        h8[e] = hh;
        l8[e] = (f16)(x - (float)hh);
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// Shared epilogue of the MFMA GEMM kernels: accumulators of a 64 x (NJ*32) wave tile -> global memory.
// `wcol_base` = first GEMM column of the wave, `mrow_base` = first row of the wave.
// Values leave the MFMA accumulators in a column-per-lane layout (2-byte scattered stores).  They are staged through
// a PER-WAVE LDS tile as fp32 (same rounding points as a direct store), 32 rows x 64 columns at a time, so that every
// lane then owns 8 consecutive columns of one row and all global traffic of the epilogue (residual, out, taps,
// split-K partials) is 16 bytes per lane.  Only wave-level ordering is needed inside (LDS executes a wave's
// instructions in order); the one block barrier separates the main loop's LDS reads from the staging writes.
// ---------------------------------------------------------------------------------------------
// phase 2 of the epilogue for one staged [nrows <= 32][NC8 * 8] fp32 tile (row stride EP_LD floats): every lane takes 8
// consecutive columns of one row per pass, 64 cells per pass in row-major cell order (NC8 = 4 / 8: 16 / 8 rows per pass, the
// original layout; NC8 = 10: the 80-column wave tile of k_gemm_p7).
// `pre` = the staged values still need bias / per-sample vector / SiLU (everything but GEGLU, which is lane-local in phase 1).
// The 16-bit residual cells of all U cells a lane owns in one staged group (U = 5 for the 32 x 80 group of k_gemm_p7, 4 / 2 for the
// 64- / 32-column groups) are requested before the first cell is worked on; the cell loop itself stays rolled (unrolled it spilled
// next to the 140 live accumulators) and picks its cell's registers by the uniform loop counter.  One cell at a time, every cell
// paid its own residual round trip (the load sits behind the tap stores, which the compiler must assume may alias it) -- with the 8
// waves of a CU all in the epilogue nothing else covered those ~1000 cycles, 20 times per tile.
template <int NC8, int EP_LD, int U_ = (32 * NC8 + 63) / 64, bool PRELOAD = true, bool X3 = false, bool XMODE = true>
__device__ __forceinline__ void epilogue_rows(const GemmParams& p, const float* stage, int mrow0, int nrows, int ocol0, int nout, int lane,
                                              int split, bool fin, bool pre) {
    constexpr int U = PRELOAD ? U_ : 1;                        // without the preload: the plain one-cell-at-a-time loop
    static_assert(U <= 5, "epilogue_rows: five residual register sets");
    const int rps = p.rows_per_sample > 0 ? p.rows_per_sample : 1;
    const int ncell = nrows * NC8;
    const bool res16 = PRELOAD && fin && p.residual && !p.res_f32;    // PRELOAD = false: kernels capped at 128 registers (4 blocks per CU)
    // XMODE = false (k_gemm_ws: 16-bit operands only, at 256 registers): the round-5 additions of the exact mode -- the fp32 residual
    // preload and the blend -- are compiled out (launch_gemm never sends such a launch there); with them, or with the other exact-mode
    // outputs removed as well, that kernel's allocation tips into scratch, which its hand-counted waits cannot see (tests/test_cabi.py)
    const bool res32 = XMODE && PRELOAD && fin && p.residual && p.res_f32;
    // fp32 residuals are taken two cells at a time (16 registers; five at a time spilled next to the live accumulators)
    const int Ueff = res32 ? (U < 2 ? U : 2) : U;
#pragma nounroll
    for (int cell0 = lane; cell0 < ncell; cell0 += 64 * Ueff) {
        bf16x8_t rr0 = {0, 0, 0, 0, 0, 0, 0, 0}, rr1 = rr0, rr2 = rr0, rr3 = rr0, rr4 = rr0;
        if (res16) {
            auto ld = [&](int u, bf16x8_t& r) {
                const int cell = cell0 + 64 * u;
                const int row = cell / NC8, c8 = (cell - row * NC8) * 8;
                const int n = ocol0 + c8, m = mrow0 + row;
                if (u < U && cell < ncell && n < nout && m < (int)p.M) r = *reinterpret_cast<const bf16x8_t*>(p.residual + (long long)m * p.ldr + n);
            };
            ld(0, rr0);
            ld(1, rr1);
            ld(2, rr2);
            ld(3, rr3);
            ld(4, rr4);
        }
        // the exact mode's fp32 residual the same way, two cells at a time (round 5): it used to be loaded inside the cell loop, one HBM
        // round trip per cell with nothing to cover it -- 17.5 of them per wave and 224 x 320 tile
        f32x4 fa0 = {0.f, 0.f, 0.f, 0.f}, fb0 = fa0, fa1 = fa0, fb1 = fa0;
        if (res32) {
            auto ld = [&](int u, f32x4& a, f32x4& b) {
                const int cell = cell0 + 64 * u;
                const int row = cell / NC8, c8 = (cell - row * NC8) * 8;
                const int n = ocol0 + c8, m = mrow0 + row;
                if (u < Ueff && cell < ncell && n < nout && m < (int)p.M) {
                    const float* rp = reinterpret_cast<const float*>(p.residual) + (long long)m * p.ldr + n;
                    a = *reinterpret_cast<const f32x4*>(rp);
                    b = *reinterpret_cast<const f32x4*>(rp + 4);
                }
            };
            ld(0, fa0, fb0);
            ld(1, fa1, fb1);
        }
#pragma nounroll
        for (int u = 0; u < Ueff; ++u) {
        const int cell = cell0 + 64 * u;
        if (cell >= ncell) continue;
        const int row = cell / NC8, c8 = (cell - row * NC8) * 8;
        const int n = ocol0 + c8;                               // output column of element 0
        if (n >= nout) continue;
        const int m = mrow0 + row;                              // rows fit 31 bits (M * ldo may not: 64-bit only in the pointer math)
        if (m >= (int)p.M) continue;
        float v[8];
        f32x4 rres0 = {0.f, 0.f, 0.f, 0.f}, rres1 = rres0;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + row * EP_LD + c8);
        const f32x4 hi4 = *reinterpret_cast<const f32x4*>(stage + row * EP_LD + c8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = lo[e];
            v[4 + e] = hi4[e];
        }
        if (!fin) {
            float* wp = p.ws + ((long long)split * p.M + m) * p.N + n;
            *reinterpret_cast<f32x4*>(wp) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(wp + 4) = f32x4{v[4], v[5], v[6], v[7]};
            continue;
        }
        if (pre) {
            if (p.bias) {                                       // 32 bytes per lane, L1-resident after the first pass
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n), b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] += b0[e];
                    v[4 + e] += b1[e];
                }
            }
            if (p.tap && p.tap_early && n < p.tap_cols) {
                f16x8 t;
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = (f16)v[e];
                *reinterpret_cast<f16x8*>(p.tap + (long long)m * p.tap_ld + n) = t;
            }
            if (p.rowvec) {
                const float* rvp = p.rowvec + (long long)(m / rps) * p.rv_stride + n;
                const f32x4 r0v = *reinterpret_cast<const f32x4*>(rvp), r1v = *reinterpret_cast<const f32x4*>(rvp + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] += r0v[e];
                    v[4 + e] += r1v[e];
                }
            }
            if (p.act == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
            }
        }
        if (p.rowadd) {
            const float ra = p.rowadd[m];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += ra;
        }
        if (p.tap && !p.tap_early && n < 2 * p.tap_cols) {
            f16x8 t;
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = (f16)v[e];
            const long long tr = tap_row(p, m) * p.tap_ld;
            if (n < p.tap_cols)
                *reinterpret_cast<f16x8*>(p.tap + tr + n) = t;
            else if (p.tap2)
                *reinterpret_cast<f16x8*>(p.tap2 + tr + (n - p.tap_cols)) = t;
        }
        if (res16) {
            bf16x8_t rr = rr0;                                  // u is uniform: scalar selects
            if (u == 1) rr = rr1;
            if (u == 2) rr = rr2;
            if (u == 3) rr = rr3;
            if (u == 4) rr = rr4;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bf16_to_f32((bf16_t)rr[e]);
        } else if (p.residual && !p.res_f32) {
            const bf16x8_t rr = *reinterpret_cast<const bf16x8_t*>(p.residual + (long long)m * p.ldr + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bf16_to_f32((bf16_t)rr[e]);
        } else if (res32) {
            f32x4 ra = fa0, rb = fb0;                           // u is uniform: scalar selects
            if (u == 1) { ra = fa1; rb = fb1; }
            rres0 = ra;
            rres1 = rb;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] += ra[e];
                v[4 + e] += rb[e];
            }
        } else if (p.residual) {
            const float* rp = reinterpret_cast<const float*>(p.residual) + (long long)m * p.ldr + n;
            const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] += r0[e];
                v[4 + e] += r1[e];
            }
        }
        if (XMODE && p.blend) {                                 // AlphaBlender on the finished value (exact VideoUNet)
            const float* bp = p.blend + (long long)m * p.ldo + n;
            f32x4 s0 = rres0, s1 = rres1;
            if (!(res32 && bp == reinterpret_cast<const float*>(p.residual) + (long long)m * p.ldr + n)) {   // VideoResBlock: blend source == skip source
                s0 = *reinterpret_cast<const f32x4*>(bp);
                s1 = *reinterpret_cast<const f32x4*>(bp + 4);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = p.blend_a * s0[e] + p.blend_b * v[e];
                v[4 + e] = p.blend_a * s1[e] + p.blend_b * v[4 + e];
            }
        }
        if (p.plane_hi && n >= p.plane_col0) {         // k | v columns of the fused projection: the attention's operand planes
            f16x8 h8, l8;
            split_cell8(v, h8, l8);
            const long long po = (long long)m * p.plane_ld + (n - p.plane_col0);
            *reinterpret_cast<f16x8*>(p.plane_hi + po) = h8;
            *reinterpret_cast<f16x8*>(p.plane_lo + po) = l8;
            continue;
        }
        const long long oo = (long long)m * p.ldo + n;
        if (p.out) {
            bf16x8_t o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (short)f32_to_bf16(v[e]);
            *reinterpret_cast<bf16x8_t*>(p.out + oo) = o;
        }
        if (p.out_f32) {
            *reinterpret_cast<f32x4*>(p.out_f32 + oo) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(p.out_f32 + oo + 4) = f32x4{v[4], v[5], v[6], v[7]};
        }
        if (p.out_split3) {                            // exact mode: the result as the next GEMM's split operand image
            f16x8 h8, l8;
            split_cell8(v, h8, l8);
            f16* o3 = p.out_split3 + (long long)m * 3 * p.ldo + n;
            *reinterpret_cast<f16x8*>(o3) = h8;
            *reinterpret_cast<f16x8*>(o3 + p.ldo) = l8;
            if (VS_THIRD_PLANE(p.ldo)) *reinterpret_cast<f16x8*>(o3 + 2 * p.ldo) = h8;
        }
        }
    }
}

// The loop over the MI x ceil(NJ/2) staged groups is ROLLED: phase 1 (accumulators -> LDS, a pure transposition except for GEGLU,
// which needs value and gate of the same lane) is selected by a uniform switch, phase 2 exists once.  Unrolled, the epilogue was
// ~40 k instructions per kernel and instruction-fetch bound (10 us per 256 x 320 tile); bias, per-sample vector and SiLU moved to
// phase 2 (same order of fp32 operations as before: + bias, + vector, SiLU, + rowadd, + residual).
template <int NJ, int MI = 2, bool PRELOAD_ = true, bool X3 = false>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16 (&acc)[MI][NJ], char* smem, long long mrow_base, int wcol_base,
                                              int lane, int wave, int split) {
    const int l31 = lane & 31, hi = lane >> 5;
    constexpr bool PRELOAD = PRELOAD_;
    constexpr int EP_LD = 68;                                  // fp32 row stride of the staging tile (64 + 4 pad)
    constexpr int NG = (NJ + 1) / 2;
    float* stage = reinterpret_cast<float*>(smem) + wave * (32 * EP_LD);
    const bool geglu = p.act == 2;
    const int nout = geglu ? p.N / 2 : p.N;
    const bool fin = p.ksplit <= 1;                            // split-K partials carry no bias/emb/activation
    __syncthreads();                                           // main-loop LDS reads are done
#pragma nounroll
    for (int ig = 0; ig < MI * NG; ++ig) {
        const int i = ig / NG, g = ig - i * NG;
        const bool two = 2 * g + 1 < NJ;
        const int wcol0 = wcol_base + g * 64;                  // first GEMM column of this 64-column group
        const int mrow0 = (int)mrow_base + i * 32;
        // ---- phase 1: accumulators -> LDS fp32 [32][ncols]
        float bx = 0.f, bg = 0.f;
        if (geglu && p.bias && (wcol0 + 32 + l31) < p.N) {
            bx = p.bias[wcol0 + l31];
            bg = p.bias[wcol0 + 32 + l31];
        }
#pragma unroll
        for (int k = 0; k < MI * NG; ++k) {
            if (ig != k) continue;                             // uniform
            constexpr int dummy = 0;
            (void)dummy;
            const int ki = k / NG, kg = k % NG;
            if (geglu) {
                if constexpr (NJ % 2 == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
                        if constexpr (X3) {                 // the exact mode's GELU: erf_f32 (common.h), the operation order of k_x_geglu_split3
                            const float gt = acc[ki][2 * kg + 1][r] + bg;
                            stage[row * EP_LD + l31] = (acc[ki][2 * kg][r] + bx) * (0.5f * gt * (1.0f + erf_f32(gt * 0.70710678118654752440f)));
                        } else {
                            stage[row * EP_LD + l31] = (acc[ki][2 * kg][r] + bx) * gelu_erf(acc[ki][2 * kg + 1][r] + bg);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    if (2 * kg + jj >= NJ) continue;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
                        stage[row * EP_LD + jj * 32 + l31] = acc[ki][(2 * kg + jj) < NJ ? (2 * kg + jj) : 0][r];
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // ---- phase 2: 8 columns per lane, coalesced 16-byte global accesses
        if (geglu || !two)
            epilogue_rows<4, EP_LD, 2, PRELOAD, X3>(p, stage, mrow0, 32, geglu ? wcol0 / 2 : wcol0, nout, lane, split, fin, !geglu);
        else
            epilogue_rows<8, EP_LD, 4, PRELOAD>(p, stage, mrow0, 32, wcol0, nout, lane, split, fin, true);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

template <int BM, int BN>
__global__ void __launch_bounds__(256, 2) k_gemm_conv(GemmParams p) {
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
    // blocks go to XCDs round-robin by blockIdx; give each XCD one contiguous chunk of the (split, tile) space, so a K split's
    // tiles (which share that split's slice of W and of A's channels) meet in one L2
    long long bid = blockIdx.x;
    {
        const long long tot = nwg * max(p.ksplit, 1);
        const long long q = tot / 8, r = tot % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int split = (int)(bid / nwg);
    bid -= split * nwg;
    // `bid` is a position in a chunk that stays on one XCD (private 4 MB L2).  Default order: N-tile fastest, so the
    // tiles resident together share activation (A) rows.  For linears whose weight matrix is far larger than L2
    // (GEGLU projections, 6-26 MB) the order is flipped inside bands of ~tiles_m/8 M-tiles: N-tile outer, M-tile
    // inner, so one or two weight tiles stay L2-resident while A streams (+4..12 % measured on those shapes).
    long long tm;
    int tn;
    if (p.ksize == 1 && (long long)p.N * p.K * 2 >= (6LL << 20) && tiles_m >= 16) {
        const long long band = (tiles_m + 7) / 8;
        const long long per_band = band * tiles_n;
        const long long bnd = bid / per_band;
        const long long inb = bid - bnd * per_band;
        const long long bw = (bnd * band + band <= tiles_m) ? band : (tiles_m - bnd * band);
        tn = (int)(inb / bw);
        tm = bnd * band + inb % bw;
    } else {
        tm = bid / tiles_n;
        tn = (int)(bid % tiles_n);
    }
    const long long m0 = tm * BM;
    const int n0 = tn * BN;

    const int Cin = p.C0 + p.C1;
    const int chunk = tid & 7;                        // 16-byte chunk inside the 128-byte K run
    const int rbase = tid >> 3;                       // 0..31

    // per-thread A row descriptors, all 32-bit: pixel-index base of the sample, top-left input coordinate of the 3x3
    // window (conv), frame index (temporal conv) or the row itself (linear).  Element offsets stay below 2^31.
    int a_base[AR], a_ih0[AR], a_iw0[AR];
    bool a_ok[AR];
    const int HWo = p.Hout * p.Wout;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const long long m = m0 + rbase + 32 * i;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? (int)m : 0;
        if (p.ksize == 1) {
            a_base[i] = mm;
            a_ih0[i] = a_iw0[i] = 0;
        } else if (p.tmode) {
            a_base[i] = mm;
            a_ih0[i] = (mm / HWo) % p.T;               // frame index t
            a_iw0[i] = 0;
        } else {
            const int b = mm / HWo, rem = mm - b * HWo;
            const int oh = rem / p.Wout, ow = rem - oh * p.Wout;
            a_base[i] = b * p.Hin * p.Win;
            a_ih0[i] = oh * p.stride - p.pad;
            a_iw0[i] = ow * p.stride - p.pad;
        }
    }
    u32x4 ra0[AR], rb0[BR];

    const int upsh = p.up - 1;                        // up is 1 or 2
    const int Hup = p.Hin << upsh, Wup = p.Win << upsh;
    const int nk_all = p.K / BK;
    const int ks_begin = p.ksplit > 1 ? (int)((long long)nk_all * split / p.ksplit) : 0;
    const int ks_end = p.ksplit > 1 ? (int)((long long)nk_all * (split + 1) / p.ksplit) : nk_all;
    int ld_k = ks_begin * BK;
    KCursor cur;
    cur.init(ld_k, p.taps, p.kchunk);
    const char* wbase = reinterpret_cast<const char*>(p.w);
    unsigned b_off[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i) b_off[i] = (unsigned)(((long long)min(n0 + rbase + 32 * i, p.N - 1) * p.K + chunk * 8) * 2);
    unsigned a_off[AR];                                // byte offset of the current (tap, source) pixel row, per A row
    bool a_val[AR];
    const char* a_src = reinterpret_cast<const char*>(p.x0);
    bool first_load = true;
    auto load_regs = [&](u32x4 (&ra)[AR], u32x4 (&rb)[BR]) {
        const int tap = cur.tap, c0 = cur.c0();        // uniform
        const int k0 = ld_k;
        // the pixel each A row reads changes only when the tap or the concat source changes (uniform branch)
        if (first_load || cur.sub == 0 || c0 == p.C0) {
            first_load = false;
            const bool second = c0 >= p.C0;
            a_src = reinterpret_cast<const char*>(second ? p.x1 : p.x0);
            const int Cs = second ? p.C1 : p.C0;
            const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                int pix;
                bool ok = a_ok[i];
                if (p.ksize == 1) {
                    pix = a_base[i];
                } else if (p.tmode) {
                    const int tt = a_ih0[i] + tap - 1;
                    ok = ok && tt >= 0 && tt < p.T;
                    pix = a_base[i] + (tap - 1) * HWo;
                } else {
                    const int ih = a_ih0[i] + kh, iw = a_iw0[i] + kw;
                    ok = ok && (unsigned)ih < (unsigned)Hup && (unsigned)iw < (unsigned)Wup;
                    pix = a_base[i] + (ih >> upsh) * p.Win + (iw >> upsh);
                }
                a_val[i] = ok;
                a_off[i] = (unsigned)(pix * Cs + chunk * 8) * 2u;
            }
        }
        const unsigned cbyte = (unsigned)fold_c(p, c0 >= p.C0 ? c0 - p.C0 : c0) * 2u;
        ld_k += BK;
        cur.advance(BK, p.taps, p.kchunk);
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (a_val[i]) v = *reinterpret_cast<const u32x4*>(a_src + (size_t)(a_off[i] + cbyte));
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (n0 + rbase + 32 * i < p.N) v = *reinterpret_cast<const u32x4*>(wbase + (size_t)(b_off[i] + (unsigned)k0 * 2u));
            rb[i] = v;
        }
    };
    auto store_lds = [&](int buf, const u32x4 (&ra)[AR], const u32x4 (&rb)[BR]) {
        char* A = smem + buf * BUF;
        char* B = A + A_BYTES;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int r = rbase + 32 * i;
            *reinterpret_cast<u32x4*>(A + r * 128 + ((chunk ^ ((r >> 1) & 7)) << 4)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int r = rbase + 32 * i;
            *reinterpret_cast<u32x4*>(B + r * 128 + ((chunk ^ ((r >> 1) & 7)) << 4)) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = ks_end - ks_begin;
    const int l31 = lane & 31, hi = lane >> 5;
    auto compute = [&](int buf) {
        const char* A = smem + buf * BUF;
        const char* B = A + A_BYTES;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8_t fa[2], fb[2];
            const int ch = s * 2 + hi;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = wm * 64 + i * 32 + l31;
                fa[i] = *reinterpret_cast<const bf16x8_t*>(A + r * 128 + ((ch ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = wn * 64 + j * 32 + l31;
                fb[j] = *reinterpret_cast<const bf16x8_t*>(B + r * 128 + ((ch ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = mfma_32x32x16(fa[i], fb[j], acc[i][j]);
        }
    };
    // Pipeline: chunk k is computed from LDS buffer k&1 while chunk k+1 is in flight into registers (issued before the
    // MFMAs, written to the other LDS buffer after them); one barrier per chunk.
    load_regs(ra0, rb0);
    store_lds(0, ra0, rb0);
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        if (ks + 1 < nk) load_regs(ra0, rb0);
        compute(ks & 1);
        if (ks + 1 < nk) store_lds((ks + 1) & 1, ra0, rb0);
        __syncthreads();
    }

    gemm_epilogue<2>(p, acc, smem, m0 + wm * 64, n0 + wn * 64, lane, wave, split);
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA variant: 128x128 block tile, BK = 32, operands go global -> LDS directly (buffer_load ... lds, 16 B per
// lane, 1 KiB per wave instruction) with the bank swizzle applied on the SOURCE side; out-of-range buffer offsets
// (padding pixels, rows >= M, weight rows >= N) land as zeros, so im2col needs no branches.  No staging VGPRs and
// 36 KiB of LDS per block -> four resident blocks (16 waves) per CU instead of two.
// LDS image of a [128][32] bf16 tile: row r at byte r*64, logical 16-byte chunk c stored at slot c ^ ((r >> 2) & 3)
// (conflict-free for ds_read_b128 MFMA fragment reads).
// ---------------------------------------------------------------------------------------------
#define DK 32
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int NST>
__global__ void __launch_bounds__(256, NST == 2 ? 4 : 3) k_gemm_dma(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE = 128 * DK * 2;               // 8 KiB per operand tile
    constexpr int BUF = 2 * TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int split, tn;
    long long tm;
    map_tile(p, 128, 128, split, tm, tn);
    const long long m0 = tm * 128;
    const int n0 = tn * 128;

    constexpr unsigned OOB = 0xF0000000u;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.w), 0, (int)((long long)p.N * p.K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x0), 0, (int)p.x0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x1 ? p.x1 : p.x0), 0, (int)p.x1_bytes, 0x00020000);

    // this lane's two A rows and two B rows (DMA pieces 2*wave and 2*wave+1, 16 rows each, 4 lanes per row)
    const int Cin = p.C0 + p.C1;
    const int HWo = p.Hout * p.Wout;
    const int upsh = p.up - 1;
    const int Hup = p.Hin << upsh, Wup = p.Win << upsh;
    int a_base[2], a_ih0[2], a_iw0[2];
    bool a_ok[2];
    unsigned a_sw[2], b_off[2];                        // swizzled chunk byte offset inside the 64-byte K run
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (2 * wave + i) * 16 + (lane >> 2);
        const unsigned csw = (unsigned)(((lane & 3) ^ ((r >> 2) & 3)) * 16);
        a_sw[i] = csw;
        const long long m = m0 + r;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? (int)m : 0;
        if (p.ksize == 1) {
            a_base[i] = mm;
            a_ih0[i] = a_iw0[i] = 0;
        } else if (p.tmode) {
            a_base[i] = mm;
            a_ih0[i] = (mm / HWo) % p.T;
            a_iw0[i] = 0;
        } else {
            const int b = mm / HWo, rem = mm - b * HWo;
            const int oh = rem / p.Wout, ow = rem - oh * p.Wout;
            a_base[i] = b * p.Hin * p.Win;
            a_ih0[i] = oh * p.stride - p.pad;
            a_iw0[i] = ow * p.stride - p.pad;
        }
        const int n = n0 + r;
        b_off[i] = n < p.N ? (unsigned)((long long)n * p.K * 2) + csw : OOB;
    }
    const int nk_all = p.K / DK;
    const int ks_begin = p.ksplit > 1 ? (int)((long long)nk_all * split / p.ksplit) : 0;
    const int ks_end = p.ksplit > 1 ? (int)((long long)nk_all * (split + 1) / p.ksplit) : nk_all;
    int ld_k = ks_begin * DK;
    KCursor cur;
    cur.init(ld_k, p.taps, p.kchunk);
    unsigned a_off[2] = {OOB, OOB};
    bool a_second = false, first = true;
    auto issue = [&](int buf) {
        const int tap = cur.tap, c0 = cur.c0(), k0 = ld_k;
        if (first || cur.sub == 0 || c0 == p.C0) {
            first = false;
            a_second = c0 >= p.C0;
            const int Cs = a_second ? p.C1 : p.C0;
            const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int pix;
                bool ok = a_ok[i];
                if (p.ksize == 1) {
                    pix = a_base[i];
                } else if (p.tmode) {
                    const int tt = a_ih0[i] + tap - 1;
                    ok = ok && tt >= 0 && tt < p.T;
                    pix = a_base[i] + (tap - 1) * HWo;
                } else {
                    const int ih = a_ih0[i] + kh, iw = a_iw0[i] + kw;
                    ok = ok && (unsigned)ih < (unsigned)Hup && (unsigned)iw < (unsigned)Wup;
                    pix = a_base[i] + (ih >> upsh) * p.Win + (iw >> upsh);
                }
                a_off[i] = ok ? (unsigned)(pix * Cs) * 2u + a_sw[i] : OOB;
            }
        }
        const unsigned cbyte = (unsigned)fold_c(p, c0 >= p.C0 ? c0 - p.C0 : c0) * 2u;
        ld_k += DK;
        cur.advance(DK, p.taps, p.kchunk);
        char* A = smem + buf * BUF;
        char* B = A + TILE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = 2 * wave + i;
            if (a_second)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x1, (lds_ptr_t)(A + piece * 1024), 16, (int)(a_off[i] + cbyte), 0, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x0, (lds_ptr_t)(A + piece * 1024), 16, (int)(a_off[i] + cbyte), 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(B + piece * 1024), 16, (int)(b_off[i] + (unsigned)k0 * 2u), 0, 0, 0);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int l31 = lane & 31, hi = lane >> 5;
    auto compute = [&](int buf) {
        const char* A = smem + buf * BUF;
        const char* B = A + TILE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8_t fa[2], fb[2];
            const int ch = s * 2 + hi;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = wm * 64 + i * 32 + l31;
                fa[i] = *reinterpret_cast<const bf16x8_t*>(A + r * 64 + ((ch ^ ((r >> 2) & 3)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = wn * 64 + j * 32 + l31;
                fb[j] = *reinterpret_cast<const bf16x8_t*>(B + r * 64 + ((ch ^ ((r >> 2) & 3)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = mfma_32x32x16(fa[i], fb[j], acc[i][j]);
        }
    };
    const int nk = ks_end - ks_begin;
    if constexpr (NST == 2) {
        issue(0);
        __syncthreads();                               // LDS-DMA outstanding -> the barrier carries s_waitcnt vmcnt(0)
        for (int ks = 0; ks + 1 < nk; ++ks) {
            issue((ks + 1) & 1);
            compute(ks & 1);
            __syncthreads();
        }
        compute((nk - 1) & 1);
    } else {
        // three LDS buffers, two K-steps in flight: wait only for the OLDER step (4 DMA instructions per wave and step),
        // raw barrier (a __syncthreads() fence would drain the younger one too), then refill the buffer freed two steps ago
        issue(0);
        if (nk > 1) issue(1);
        int cb = 0, ib = 2;                            // compute / issue buffer indices (mod 3)
        for (int ks = 0; ks < nk; ++ks) {
            if (ks + 1 < nk)
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (ks + 2 < nk) issue(ib);
            compute(cb);
            cb = cb == 2 ? 0 : cb + 1;
            ib = ib == 2 ? 0 : ib + 1;
        }
    }
    gemm_epilogue<2, 2, NST != 2>(p, acc, smem, m0 + wm * 64, n0 + wn * 64, lane, wave, split);
}

// ---------------------------------------------------------------------------------------------
// Wide-tile LDS-DMA kernels: (WM*64) x (NJ*64) block tile, WM x 2 waves, wave tile 64 x (NJ*32).
//   <NJ, 4, 64>  "big": 256 x 256 / 256 x 320, BK = 64, 8 waves, one block per CU -- long-K convolutions.
//   <NJ, 4, 32, 1> "mid": 128 x 256 / 128 x 320, BK = 32, 8 waves (wave tile 32 x 160), two blocks per CU: A is read once per
//                row tile instead of once per 128 columns.  Wins only on 28672 x 640 x 640 (-17 %); a 4-wave variant of the
//                same tile was 1.1-2x slower everywhere (half the resident waves).
// NJ = 5 covers N = 320/640/960/1280/1920 without padding waste.  The 128x128 kernels read 1 byte of operand through
// the CU's vector-memory path (64 B/clk) per 64 MFMA flops -- exactly the MFMA rate, so they cannot pass ~50 % of
// peak; these tiles read 0.45-0.7 bytes per 64 flops.
// LDS image per operand tile: [rows][BK] bf16; 16-byte chunk c of row r is stored at slot c ^ ((r>>1)&7) for 128-byte
// rows and c ^ ((r>>2)&3) for 64-byte rows (conflict-free ds_read_b128 fragment reads).
// ---------------------------------------------------------------------------------------------
template <int BKT>
__device__ __forceinline__ int lds_swz(int r) {
    return BKT == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3);
}

template <int NJ, int WM, int BKT, int MI = 2>
__global__ void __launch_bounds__(WM * 128, MI == 1 ? 4 : (WM == 4 ? 1 : 2)) k_gemm_tile(GemmParams p) {   // 2nd = min waves per SIMD
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = WM * MI * 32, BN = NJ * 64, NW = WM * 2;
    constexpr int RB = BKT * 2;                            // bytes per LDS row
    constexpr int RPP = 1024 / RB;                         // rows per 1-KiB DMA piece
    constexpr int CPR = RB / 16;                           // 16-byte chunks per row
    constexpr int APIECES = BM / RPP, BPIECES = BN / RPP;
    constexpr int APW = (APIECES + NW - 1) / NW, BPW = (BPIECES + NW - 1) / NW;   // pieces per wave (piece = wave + i*NW)
    constexpr int A_BYTES = BM * RB, B_BYTES = BN * RB, BUF = A_BYTES + B_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int split, tn;
    long long tm;
    map_tile(p, BM, BN, split, tm, tn);
    const long long m0 = tm * BM;
    const int n0 = tn * BN;

    constexpr unsigned OOB = 0xF0000000u;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.w), 0, (int)((long long)p.N * p.K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x0), 0, (int)p.x0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x1 ? p.x1 : p.x0), 0, (int)p.x1_bytes, 0x00020000);

    // wave w stages A pieces w, w+NW, .. and B pieces w, w+NW, ..; lane l of a piece: row l / CPR, chunk l % CPR
    const int Cin = p.C0 + p.C1;
    const int HWo = p.Hout * p.Wout;
    const int upsh = p.up - 1;
    const int Hup = p.Hin << upsh, Wup = p.Win << upsh;
    const int lrow = lane / CPR, lch = lane % CPR;
    int a_base[APW], a_ih0[APW], a_iw0[APW];
    bool a_ok[APW];
#pragma unroll
    for (int i = 0; i < APW; ++i) {
        const int r = (wave + i * NW) * RPP + lrow;
        const long long m = m0 + r;
        a_ok[i] = m < p.M && (wave + i * NW) < APIECES;
        const int mm = a_ok[i] ? (int)m : 0;
        if (p.ksize == 1) {
            a_base[i] = mm;
            a_ih0[i] = a_iw0[i] = 0;
        } else if (p.tmode) {
            a_base[i] = mm;
            a_ih0[i] = (mm / HWo) % p.T;
            a_iw0[i] = 0;
        } else {
            const int b = mm / HWo, rem = mm - b * HWo;
            const int oh = rem / p.Wout, ow = rem - oh * p.Wout;
            a_base[i] = b * p.Hin * p.Win;
            a_ih0[i] = oh * p.stride - p.pad;
            a_iw0[i] = ow * p.stride - p.pad;
        }
    }
    unsigned b_off[BPW];
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
        const int r = (wave + i * NW) * RPP + lrow;
        const int n = n0 + r;
        b_off[i] = (n < p.N && (wave + i * NW) < BPIECES) ? (unsigned)((long long)n * p.K * 2) + (unsigned)((lch ^ lds_swz<BKT>(r)) * 16) : OOB;
    }
    const int nk_all = p.K / BKT;
    const int ks_begin = p.ksplit > 1 ? (int)((long long)nk_all * split / p.ksplit) : 0;
    const int ks_end = p.ksplit > 1 ? (int)((long long)nk_all * (split + 1) / p.ksplit) : nk_all;
    int ld_k = ks_begin * BKT;
    KCursor cur;
    cur.init(ld_k, p.taps, p.kchunk);
    unsigned a_off[APW];
#pragma unroll
    for (int i = 0; i < APW; ++i) a_off[i] = OOB;
    bool a_second = false, first = true;
    auto issue = [&](int buf) {
        const int tap = cur.tap, c0 = cur.c0(), k0 = ld_k;
        if (first || cur.sub == 0 || c0 == p.C0) {
            first = false;
            a_second = c0 >= p.C0;
            const int Cs = a_second ? p.C1 : p.C0;
            const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
            for (int i = 0; i < APW; ++i) {
                int pix;
                bool ok = a_ok[i];
                if (p.ksize == 1) {
                    pix = a_base[i];
                } else if (p.tmode) {
                    const int tt = a_ih0[i] + tap - 1;
                    ok = ok && tt >= 0 && tt < p.T;
                    pix = a_base[i] + (tap - 1) * HWo;
                } else {
                    const int ih = a_ih0[i] + kh, iw = a_iw0[i] + kw;
                    ok = ok && (unsigned)ih < (unsigned)Hup && (unsigned)iw < (unsigned)Wup;
                    pix = a_base[i] + (ih >> upsh) * p.Win + (iw >> upsh);
                }
                const int r = (wave + i * NW) * RPP + lrow;
                a_off[i] = ok ? (unsigned)(pix * Cs) * 2u + (unsigned)((lch ^ lds_swz<BKT>(r)) * 16) : OOB;
            }
        }
        const unsigned cbyte = (unsigned)fold_c(p, c0 >= p.C0 ? c0 - p.C0 : c0) * 2u;
        ld_k += BKT;
        cur.advance(BKT, p.taps, p.kchunk);
        char* A = smem + buf * BUF + wave * 1024;
        char* B = smem + buf * BUF + A_BYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            if (APIECES % NW != 0 && wave + i * NW >= APIECES) break;           // wave-uniform
            if (a_second)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x1, (lds_ptr_t)(A + i * (NW * 1024)), 16, (int)(a_off[i] + cbyte), 0, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x0, (lds_ptr_t)(A + i * (NW * 1024)), 16, (int)(a_off[i] + cbyte), 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < BPW; ++i) {
            if (BPIECES % NW != 0 && wave + i * NW >= BPIECES) break;           // wave-uniform
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(B + i * (NW * 1024)), 16, (int)(b_off[i] + (unsigned)k0 * 2u), 0, 0, 0);
        }
    };

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int l31 = lane & 31, hi = lane >> 5;
    const int sw = lds_swz<BKT>(l31);                          // every fragment row is l31 + a multiple of 32
    const int arow = (wm * (MI * 32) + l31) * RB, brow = (wn * (NJ * 32) + l31) * RB;
    auto compute = [&](int buf) {
        const char* A = smem + buf * BUF + arow;
        const char* B = smem + buf * BUF + A_BYTES + brow;
#pragma unroll
        for (int s = 0; s < BKT / 16; ++s) {
            bf16x8_t fa[MI];
            const int co = ((s * 2 + hi) ^ sw) << 4;
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(A + i * 32 * RB + co);
            if constexpr (MI == 1) {                               // 128-VGPR budget: one B fragment live at a time
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const bf16x8_t fb = *reinterpret_cast<const bf16x8_t*>(B + j * 32 * RB + co);
                    acc[0][j] = mfma_32x32x16(fa[0], fb, acc[0][j]);
                }
            } else {
                bf16x8_t fb[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(B + j * 32 * RB + co);
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int i = 0; i < MI; ++i)
                        acc[i][j] = mfma_32x32x16(fa[i], fb[j], acc[i][j]);
            }
        }
    };
    const int nk = ks_end - ks_begin;
    issue(0);
    __syncthreads();
    for (int ks = 0; ks + 1 < nk; ++ks) {
        issue((ks + 1) & 1);
        compute(ks & 1);
        __syncthreads();
    }
    compute((nk - 1) & 1);
    gemm_epilogue<NJ, MI, MI != 1>(p, acc, smem, m0 + wm * (MI * 32), n0 + wn * (NJ * 32), lane, wave, split);
}

// ---------------------------------------------------------------------------------------------
// Phased big tile: the same 256 x (NJ*64) block / 64 x (NJ*32) wave tile as k_gemm_tile<NJ,4,64>, but the K-tile is worked
// off in NJ phases (phase j = column block j of the wave tile over the full BK = 64: 8 MFMAs, 256 cycles) and the two halves
// of the workgroup (waves 0-3 / 4-7, one of each per SIMD) run one barrier apart, so while one half feeds the matrix
// pipe the other reads LDS and issues the LDS-DMA of a later tile.  Two LDS buffers, restaged region by region:
//   group G0(u) = A rows 0-127 + B block 0, G1(u) = A rows 128-255 + B block 1, Gg(u) = B block g  (tile u)
//   Gg(u) is issued in the read section of phase (u-2, g+2) [A and B block g of tile u-2 were last read in phases (u-2, 0) and
//   (u-2, g); two phases later every wave of both halves has retired those reads], i.e. 5-7 phases before its first use.
// Waits are counted (loads retire in order): before phase q's first barrier each wave waits until at most VM(q+1) of its own
// loads are outstanding, which covers everything phase q+1 reads; the barrier publishes it to the other waves.  Tiles beyond the
// K range are staged with out-of-range buffer offsets (zeros, no memory traffic) so the counts stay uniform.
// ---------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int NJ, bool X3 = false>
__global__ void __launch_bounds__(512, 2) k_gemm_ph(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 256, BN = NJ * 64, RB = 128;
    constexpr int A_BYTES = BM * RB, B_BYTES = BN * RB;              // LDS map: A[0] A[1] B[0] B[1] (ds offsets stay within 16 bits)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // scalar: DMA destinations and piece rows live in SGPRs
    const int wm = wave >> 1, wn = wave & 1, grp = wave >> 2;
    int split, tn;
    long long tm;
    map_tile(p, BM, BN, split, tm, tn);
    const long long m0 = tm * BM;
    const int n0 = tn * BN;

    constexpr unsigned OOB = 0xF0000000u;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.w), 0, (int)((long long)p.N * p.K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x0), 0, (int)p.x0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x1 ? p.x1 : p.x0), 0, (int)p.x1_bytes, 0x00020000);

    // staging: a DMA piece is 8 rows x 128 B; lane l -> row l/8, 16-byte chunk l%8.  A set s4 = (half h = s4>>1, i = s4&1) is
    // piece wave + 8i of rows h*128..; B block g is 64 rows (32 of each column half of the wave grid), piece = wave.  Every
    // staged row of this lane has the same (row>>1)&7, so one swizzled chunk offset serves all of them.
    const int Cin = p.C0 + p.C1;
    const int HWo = p.Hout * p.Wout;
    const int upsh = p.up - 1;
    const int Hup = p.Hin << upsh, Wup = p.Win << upsh;
    const int lrow = lane >> 3, lch = lane & 7;
    const unsigned swz16 = (unsigned)((lch ^ (((wave & 1) << 2) | (lrow >> 1))) << 4);
    // One branch-free address form for the three A operands: pixel = base + ((ih0 + kh) >> upsh) * wmul + ((iw0 + kw) >> upsh),
    // valid iff 0 <= ih0 + kh < hb and 0 <= iw0 + kw < wb.  3x3 conv: (kh, kw) = tap; temporal 3-tap conv: the "row" is the frame
    // index (kh = tap, wmul = Hout*Wout, hb = T); linear: one tap, ih0 = iw0 = 0.  Rows past M carry ih0 = -0x4000.
    const int hb = p.ksize == 1 ? 1 : (p.tmode ? p.T : Hup), wb = (p.ksize == 1 || p.tmode) ? 1 : Wup;
    const int wmul = p.tmode ? HWo : p.Win;
    int a_base[4], a_hw[4];                                            // a_hw = (ih0 + 0x4000) << 16 | (iw0 + 0x4000)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const int r = (s4 >> 1) * 128 + (wave + 8 * (s4 & 1)) * 8 + lrow;
        const long long m = m0 + r;
        const bool ok = m < p.M;
        const int mm = ok ? (int)m : 0;
        int ih0 = 0, iw0 = 0;
        if (p.ksize == 1) {
            a_base[s4] = mm;
        } else if (p.tmode) {
            const int t = (mm / HWo) % p.T;
            a_base[s4] = mm - t * HWo;
            ih0 = t - 1;
        } else {
            const int b = mm / HWo, rem = mm - b * HWo;
            const int oh = rem / p.Wout, ow = rem - oh * p.Wout;
            a_base[s4] = b * p.Hin * p.Win;
            ih0 = oh * p.stride - p.pad;
            iw0 = ow * p.stride - p.pad;
        }
        if (!ok) ih0 = -0x4000;
        a_hw[s4] = ((ih0 + 0x4000) << 16) | (iw0 + 0x4000);
    }
    const int b_r0 = (wave >> 2) * (NJ * 32) + (wave & 3) * 8;              // first row of this wave's piece inside B block 0
    const int b_n = n0 + b_r0 + lrow;                                        // + g*32: the weight row this lane stages for block g
    const unsigned b_off0 = (unsigned)((long long)b_n * p.K * 2) + swz16;
    const unsigned b_gstep = (unsigned)p.K * 64u;                            // 32 weight rows
    const int nk_all = p.K / 64;
    const int ks_begin = p.ksplit > 1 ? (int)((long long)nk_all * split / p.ksplit) : 0;
    const int ks_end = p.ksplit > 1 ? (int)((long long)nk_all * (split + 1) / p.ksplit) : nk_all;
    const int nk = ks_end - ks_begin;
    // A addressing: with the chunk-major K order the tap changes every K-tile, so the two rows a stage call loads are addressed
    // from (a_base, a_hw) on the spot (about 10 VALU ops per row, inside the read section that runs under the other half's MFMAs);
    // the tile-uniform part (tap -> kh, kw; source; channel offset) is scalar
    KCursor cur;
    cur.init(ks_begin * 64, p.taps, p.kchunk);
    int t_kh = 0, t_kw = 0, t_Cs = 0, t_cc = 0;
    bool a_second = false;
    // stage group g of tile u into buffer buf (groups of one tile are staged in order g = 0, 1, ..; the cursor moves after G1)
    auto stage = [&](int g, int u, int buf) {
        const bool live = u < nk;
        if (g == 0) {
            const int c0 = cur.c0();
            a_second = c0 >= p.C0;
            t_Cs = a_second ? p.C1 : p.C0;
            t_cc = fold_c(p, a_second ? c0 - p.C0 : c0);
            const int t3 = cur.tap / 3;
            t_kh = p.tmode ? cur.tap : t3;
            t_kw = p.tmode ? 0 : cur.tap - t3 * 3;
            cur.advance(64, p.taps, p.kchunk);
        }
        if (g < 2) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int s4 = g * 2 + i;
                char* dst = smem + buf * A_BYTES + (g * 128 + (wave + 8 * i) * 8) * RB;
                const int ih = (a_hw[s4] >> 16) - 0x4000 + t_kh, iw = (a_hw[s4] & 0xFFFF) - 0x4000 + t_kw;
                const bool ok = live && (unsigned)ih < (unsigned)hb && (unsigned)iw < (unsigned)wb;
                const int pix = a_base[s4] + (ih >> upsh) * wmul + (iw >> upsh);
                const unsigned off = ok ? (unsigned)(pix * t_Cs + t_cc) * 2u + swz16 : OOB;
                if (a_second)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x1, (lds_ptr_t)dst, 16, (int)off, 0, 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x0, (lds_ptr_t)dst, 16, (int)off, 0, 0, 0);
            }
        }
        char* dst = smem + 2 * A_BYTES + buf * B_BYTES + (b_r0 + g * 32) * RB;
        const unsigned off = (live && b_n + g * 32 < p.N) ? b_off0 + (unsigned)g * b_gstep + (unsigned)(ks_begin + u) * 128u : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)dst, 16, (int)off, 0, 0, 0);
    };

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int l31 = lane & 31, hi = lane >> 5;
    const int sw = lds_swz<64>(l31);
    const int arow = (wm * 64 + l31) * RB, brow = 2 * A_BYTES + (wn * (NJ * 32) + l31) * RB;

    // prologue: everything the steady state would have issued before phase (0, 0)
#pragma unroll
    for (int g = 0; g < NJ; ++g) stage(g, 0, 0);
#pragma unroll
    for (int g = 0; g + 2 < NJ; ++g) stage(g, 1, 1);
    constexpr int CNT_ALL = 2 * (NJ + 4);
    // loads of the groups staged at phase positions j', j'+1, j'+2 (and 0..3 for j' = 0) are the only ones that may still fly
    constexpr auto cnt = [](int pos) { return (pos == 2 || pos == 3) ? 3 : 1; };
    wait_vmcnt<CNT_ALL - 8>();
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();

    bf16x8_t fa[2][4];
    auto tile = [&](int t, int buf) {
        const char* A = smem + buf * A_BYTES + arow;
        const char* B = smem + buf * B_BYTES + brow;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            // ---- read section
            bf16x8_t fb[4];
            if (j == 0) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < 2; ++i) fa[i][s] = *reinterpret_cast<const bf16x8_t*>(A + i * 32 * RB + (((s * 2 + hi) ^ sw) << 4));
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) fb[s] = *reinterpret_cast<const bf16x8_t*>(B + j * 32 * RB + (((s * 2 + hi) ^ sw) << 4));
            if (j >= 2)
                stage(j - 2, t + 2, buf);
            else
                stage(j - 2 + NJ, t + 1, buf ^ 1);
            // counted wait for what phase j+1 reads
            const int jn = (j + 1) % NJ;
            if (jn == 0)
                wait_vmcnt<CNT_ALL - 8>();
            else if (cnt(jn) + cnt((jn + 1) % NJ) + cnt((jn + 2) % NJ) == 7)
                wait_vmcnt<CNT_ALL - 7>();
            else if (cnt(jn) + cnt((jn + 1) % NJ) + cnt((jn + 2) % NJ) == 5)
                wait_vmcnt<CNT_ALL - 5>();
            else
                wait_vmcnt<CNT_ALL - 3>();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- matrix section
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i][j] = mfma_32x32x16(fa[i][s], fb[s], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int t = 0; t < nk; t += 2) {
        tile(t, 0);
        if (t + 1 < nk) tile(t + 1, 1);
    }
    wait_vmcnt<0>();
    if (grp == 0) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
    gemm_epilogue<NJ, 2, true, X3>(p, acc, smem, m0 + wm * 64, n0 + wn * (NJ * 32), lane, wave, split);
}

// ---------------------------------------------------------------------------------------------
// k_gemm_p7: the phased big tile for M = 7 * 2^k * 32.  A CFG window has 2 * 14 = 28 = 4 * 7 samples, so every GEMM of the
// UNet has M = 7 * 2^k rows and any power-of-two tile height leaves the grid at 7/8 of a round of 256 CUs (448 tiles of
// 256 rows at the 64x64 level, 224 at 32x32, 112 / 28 before split-K below).  This tile is 224 x 320: 512 / 256 / 128 / 32
// tiles on the same layers -- whole rounds.  224 = 7 * 32 cannot be split over 8 waves in 32-row fragments, so the
// fragments are v_mfma_f32_16x16x32: wave grid 2 (M) x 4 (N), wave tile 112 x 80 = 7 x 5 fragments (35 accumulators of 4
// registers; 7 A + 5 B fragment reads per 35 MFMAs of 16 cycles -- the same LDS bytes per flop as the 64 x 160 wave tile).
// Staging pieces, LDS image and swizzle are those of k_gemm_ph<5> (the staging SCHEDULE is its own, see the kernel): the A region
// keeps 256 rows and the rows 224..255 of a tile are staged with out-of-range offsets (zeros, no memory traffic), so every
// wave still issues 9 DMA instructions per K-tile.  A B block g is now the four 16-row strips {wn * 80 + g * 16 ..} that
// phase g reads.  Phase j = B fragment column j over the whole BK = 64: 14 MFMAs (7 A fragments x 2 k-steps, 224 cycles).
// ---------------------------------------------------------------------------------------------
// Phase 2 of the GEGLU projection on 16-wide fragments (k_gemm_p7x<4, true>): fragment columns (2 jj, 2 jj + 1) of a wave tile hold value
// and gate of the same 16 output columns (weight rows interleaved in 16-row value | gate groups, exact.pack_geglu_x), staged side by
// side.  A lane takes 8 consecutive product columns of one row per pass: (value + b) * gelu(gate + b) with erf_f32 (common.h) in the operation
// order of k_x_geglu_split3, split into (hi, lo) and written as the consumer's operand image [hi | lo | hi].  A rolled loop of 8 erf
// per pass: formed while staging (16 inlined erf per fragment pair, unrolled over the groups) it spilled into scratch.
template <int NJ, int EP_LD>
__device__ __forceinline__ void epilogue_rows_geglu16(const GemmParams& p, const float* stage, int mrow0, int nrows, int wcol_base, int lane) {
    constexpr int NC8 = NJ;                                    // NJ / 2 pairs x 16 product columns = NJ cells of 8
    const int ncell = nrows * NC8;
#pragma nounroll
    for (int cell = lane; cell < ncell; cell += 64) {
        const int row = cell / NC8, pc = (cell - row * NC8) * 8;  // first product column of the cell inside the wave tile
        const int jj = pc >> 4, c = pc & 15;
        const int nv = wcol_base + jj * 32 + c;                 // GEMM column of the first value; its gate sits 16 columns further
        const int m = mrow0 + row;
        if (m >= (int)p.M || nv + 24 > p.N) continue;            // value columns nv .. nv + 7, gate columns nv + 16 .. nv + 23
        const float* sv = stage + row * EP_LD + jj * 32 + c;
        f32x4 v0 = *reinterpret_cast<const f32x4*>(sv), v1 = *reinterpret_cast<const f32x4*>(sv + 4);
        f32x4 g0 = *reinterpret_cast<const f32x4*>(sv + 16), g1 = *reinterpret_cast<const f32x4*>(sv + 20);
        if (p.bias) {
            const f32x4 bv0 = *reinterpret_cast<const f32x4*>(p.bias + nv), bv1 = *reinterpret_cast<const f32x4*>(p.bias + nv + 4);
            const f32x4 bg0 = *reinterpret_cast<const f32x4*>(p.bias + nv + 16), bg1 = *reinterpret_cast<const f32x4*>(p.bias + nv + 20);
            v0 += bv0;
            v1 += bv1;
            g0 += bg0;
            g1 += bg1;
        }
        f16x8 h8, l8;
        float pr[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float gt = e < 4 ? g0[e & 3] : g1[e & 3];
            pr[e] = (e < 4 ? v0[e & 3] : v1[e & 3]) * (0.5f * gt * (1.0f + erf_f32(gt * 0.70710678118654752440f)));
        }
        split_cell8(pr, h8, l8);                                // one fp32 value for both planes (common.h: split_hl)
        f16* o3 = p.out_split3 + (long long)m * 3 * p.ldo + (wcol_base >> 1) + pc;
        *reinterpret_cast<f16x8*>(o3) = h8;
        *reinterpret_cast<f16x8*>(o3 + p.ldo) = l8;
        if (VS_THIRD_PLANE(p.ldo)) *reinterpret_cast<f16x8*>(o3 + 2 * p.ldo) = h8;
    }
}

template <int MI, int NJ, bool G3 = false>
__device__ __forceinline__ void gemm_epilogue16(const GemmParams& p, f32x4 (&acc)[MI][NJ], char* smem, int mrow_base, int wcol_base, int lane,
                                                int wave, int split) {
    constexpr int EP_LD = NJ * 16 + 4;                         // fp32 row stride of the staging tile
    constexpr int NG = (MI + 1) / 2;                           // 32-row groups (two 16-row fragments each)
    float* stage = reinterpret_cast<float*>(smem) + wave * (32 * EP_LD);
    const int l15 = lane & 15, q = lane >> 4;
    const bool fin = p.ksplit <= 1;                            // split-K partials carry no bias/emb/activation
    static_assert(!G3 || NJ % 2 == 0, "GEGLU pairs fragment columns");
    __syncthreads();                                           // main-loop LDS reads are done
#pragma nounroll
    for (int ig = 0; ig < NG; ++ig) {
        const int nrows = (2 * ig + 1 < MI) ? 32 : 16;
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            if (ig != k) continue;                             // uniform: accumulator indices stay compile-time
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                if (2 * k + ii >= MI) continue;
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        stage[(ii * 16 + q * 4 + r) * EP_LD + j * 16 + l15] = acc[(2 * k + ii) < MI ? (2 * k + ii) : 0][j][r];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if constexpr (G3)
            epilogue_rows_geglu16<NJ, EP_LD>(p, stage, mrow_base + ig * 32, nrows, wcol_base, lane);
        else
            epilogue_rows<NJ * 2, EP_LD>(p, stage, mrow_base + ig * 32, nrows, wcol_base, p.N, lane, split, fin, true);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

template <int NPH>
__global__ void __launch_bounds__(512, 2) k_gemm_p7(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NJ = 5, MI = 7;
    constexpr int BM = 224, BN = 320, RB = 128;
    constexpr int A_BYTES = 256 * RB, B_BYTES = BN * RB;                 // LDS map: A[0] A[1] B[0] B[1]; A keeps 256 rows (see above)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, grp = wave >> 2;
    int split, tn;
    long long tm;
    map_tile(p, BM, BN, split, tm, tn);
    const long long m0 = tm * BM;
    const int n0 = tn * BN;

    constexpr unsigned OOB = 0xF0000000u;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.w), 0, (int)((long long)p.N * p.K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x0), 0, (int)p.x0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x1 ? p.x1 : p.x0), 0, (int)p.x1_bytes, 0x00020000);

    const int HWo = p.Hout * p.Wout;
    const int upsh = p.up - 1;
    const int Hup = p.Hin << upsh, Wup = p.Win << upsh;
    const int lrow = lane >> 3, lch = lane & 7;
    const unsigned swz16 = (unsigned)((lch ^ (((wave & 1) << 2) | (lrow >> 1))) << 4);
    const int hb = p.ksize == 1 ? 1 : (p.tmode ? p.T : Hup), wb = (p.ksize == 1 || p.tmode) ? 1 : Wup;
    const int wmul = p.tmode ? HWo : p.Win;
    int a_base[4], a_hw[4];                                            // a_hw = (ih0 + 0x4000) << 16 | (iw0 + 0x4000)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const int r = (s4 >> 1) * 128 + (wave + 8 * (s4 & 1)) * 8 + lrow;
        const long long m = m0 + r;
        const bool ok = r < BM && m < p.M;                             // rows 224..255 of the LDS image belong to no tile
        const int mm = ok ? (int)m : 0;
        int ih0 = 0, iw0 = 0;
        if (p.ksize == 1) {
            a_base[s4] = mm;
        } else if (p.tmode) {
            const int t = (mm / HWo) % p.T;
            a_base[s4] = mm - t * HWo;
            ih0 = t - 1;
        } else {
            const int b = mm / HWo, rem = mm - b * HWo;
            const int oh = rem / p.Wout, ow = rem - oh * p.Wout;
            a_base[s4] = b * p.Hin * p.Win;
            ih0 = oh * p.stride - p.pad;
            iw0 = ow * p.stride - p.pad;
        }
        if (!ok) ih0 = -0x4000;
        a_hw[s4] = ((ih0 + 0x4000) << 16) | (iw0 + 0x4000);
    }
    // B block g = the 16-row strips wn * 80 + g * 16 .. of the four wave columns; piece `wave` = strip wave >> 1, rows (wave & 1) * 8 ..
    const int b_r0 = (wave >> 1) * 80 + (wave & 1) * 8;
    const int b_n = n0 + b_r0 + lrow;                                        // + g*16: the weight row this lane stages for block g
    const unsigned b_off0 = (unsigned)((long long)b_n * p.K * 2) + swz16;
    const unsigned b_gstep = (unsigned)p.K * 32u;                            // 16 weight rows
    const int nk_all = p.K / 64;
    const int ks_begin = p.ksplit > 1 ? (int)((long long)nk_all * split / p.ksplit) : 0;
    const int ks_end = p.ksplit > 1 ? (int)((long long)nk_all * (split + 1) / p.ksplit) : nk_all;
    const int nk = ks_end - ks_begin;
    KCursor cur;
    cur.init(ks_begin * 64, p.taps, p.kchunk);
    int t_kh = 0, t_kw = 0, t_Cs = 0, t_cc = 0;
    bool a_second = false;
    // One A set (s4: rows (s4 >> 1) * 128 + (wave + 8 * (s4 & 1)) * 8 .., one DMA instruction per wave) / one B block of tile u into
    // buffer buf.  The K cursor moves when set 0 of a tile is staged; the sets of one tile are staged in order 0, 1, 2, 3.
    auto stage_a = [&](int s4, int u, int buf) {
        const bool live = u < nk;
        if (s4 == 0) {
            const int c0 = cur.c0();
            a_second = c0 >= p.C0;
            t_Cs = a_second ? p.C1 : p.C0;
            t_cc = fold_c(p, a_second ? c0 - p.C0 : c0);
            const int t3 = cur.tap / 3;
            t_kh = p.tmode ? cur.tap : t3;
            t_kw = p.tmode ? 0 : cur.tap - t3 * 3;
            cur.advance(64, p.taps, p.kchunk);
        }
        char* dst = smem + buf * A_BYTES + ((s4 >> 1) * 128 + (wave + 8 * (s4 & 1)) * 8) * RB;
        const int ih = (a_hw[s4] >> 16) - 0x4000 + t_kh, iw = (a_hw[s4] & 0xFFFF) - 0x4000 + t_kw;
        const bool ok = live && (unsigned)ih < (unsigned)hb && (unsigned)iw < (unsigned)wb;
        const int pix = a_base[s4] + (ih >> upsh) * wmul + (iw >> upsh);
        const unsigned off = ok ? (unsigned)(pix * t_Cs + t_cc) * 2u + swz16 : OOB;
        if (a_second)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x1, (lds_ptr_t)dst, 16, (int)off, 0, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x0, (lds_ptr_t)dst, 16, (int)off, 0, 0, 0);
    };
    auto stage_b = [&](int g, int u, int buf) {
        const bool live = u < nk;
        char* dst = smem + 2 * A_BYTES + buf * B_BYTES + (b_r0 + g * 16) * RB;
        const unsigned off = (live && b_n + g * 16 < p.N) ? b_off0 + (unsigned)g * b_gstep + (unsigned)(ks_begin + u) * 128u : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)dst, 16, (int)off, 0, 0, 0);
    };

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int sw = (l15 >> 1) & 7;                                    // every fragment row is l15 + a multiple of 16
    const int arow = (wm * 112 + l15) * RB, brow = 2 * A_BYTES + (wn * 80 + l15) * RB;

    // Staging schedule: one A set + one B block per phase -- phases 2, 3, 4 of tile t stage (A set j-2, B block j-2) of tile t+2 into
    // tile t's buffer (its A rows were last read in phase 0, its B block j-2 in phase j-2: two phases earlier), phase 0 stages
    // (A set 3, B block 3) and phase 1 B block 4 of tile t+1: 2,2,2,2,1 DMA instructions per wave and phase (k_gemm_ph's groups are
    // 3,3,1,1,1; an LDS-DMA instruction costs its wave 100-185 cycles inside a read section, MI355X_MICROARCH.md; measured -1 %).
    // Per-wave issue order of a tile: A0 B0 A1 B1 A2 B2 A3 B3 B4.
    bf16x8_t fa[MI][2];
    if constexpr (NPH == 3) {
        // Three phases per K-tile -- B fragment columns {0,1}, {2,3}, {4}: 28 / 28 / 14 MFMAs between barriers, 6 barriers per K-tile
        // instead of 10.  Staging (three DMA instructions per wave and phase): phase 0 of tile t stages A3, B0, B1 of tile t+1, phase 1
        // B2, B3, B4 of tile t+1 (both into tile t-1's buffer: B4 was last read two phases earlier), phase 2 A0, A1, A2 of tile t+2
        // (into tile t's buffer: its A rows were last read in phase 0).  Per-wave issue order of a tile: A0 A1 A2 | A3 B0 B1 | B2 B3 B4.
        // Counted waits (younger loads than the last one the next phase needs): before phase 1: B4 + 3 + 3 = 7, before phase 2: 9,
        // before the next tile's phase 0: 6.
#pragma unroll
        for (int g = 0; g < 4; ++g) stage_a(g, 0, 0);
#pragma unroll
        for (int g = 0; g < NJ; ++g) stage_b(g, 0, 0);
#pragma unroll
        for (int g = 0; g < 3; ++g) stage_a(g, 1, 1);
        wait_vmcnt<6>();
        __builtin_amdgcn_s_barrier();
        if (grp == 1) __builtin_amdgcn_s_barrier();
        auto tile3 = [&](int t, int buf) {
            const char* A = smem + buf * A_BYTES + arow;
            const char* B = smem + buf * B_BYTES + brow;
#pragma unroll
            for (int ph = 0; ph < 3; ++ph) {
                const int j0 = 2 * ph, nj = ph == 2 ? 1 : 2;
                bf16x8_t fb[2][2];
                if (ph == 0) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) fa[i][0] = *reinterpret_cast<const bf16x8_t*>(A + i * 16 * RB + ((l4 ^ sw) << 4));
                }
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
                        if (jj < nj) fb[jj][kk] = *reinterpret_cast<const bf16x8_t*>(B + (j0 + jj) * 16 * RB + (((kk * 4 + l4) ^ sw) << 4));
                if (ph == 0) {
                    stage_a(3, t + 1, buf ^ 1);
                    stage_b(0, t + 1, buf ^ 1);
                    stage_b(1, t + 1, buf ^ 1);
                    wait_vmcnt<7>();
                } else if (ph == 1) {
                    stage_b(2, t + 1, buf ^ 1);
                    stage_b(3, t + 1, buf ^ 1);
                    stage_b(4, t + 1, buf ^ 1);
                    wait_vmcnt<9>();
                } else {
                    stage_a(0, t + 2, buf);
                    stage_a(1, t + 2, buf);
                    stage_a(2, t + 2, buf);
                    wait_vmcnt<6>();
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (ph == 0) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) fa[i][1] = *reinterpret_cast<const bf16x8_t*>(A + i * 16 * RB + (((4 + l4) ^ sw) << 4));
                }
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                        for (int i = 0; i < MI; ++i)
                            if (jj < nj) acc[i][j0 + jj] = mfma_16x16x32(fa[i][kk], fb[jj][kk], acc[i][j0 + jj]);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        for (int t = 0; t < nk; t += 2) {
            tile3(t, 0);
            if (t + 1 < nk) tile3(t + 1, 1);
        }
    } else {
    // prologue: everything the steady state would have issued before phase (0, 0); that phase needs A0..A3 and B0 of tile 0, so B3,
    // B4 and tile 1's six loads may still fly
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        stage_a(g, 0, 0);
        stage_b(g, 0, 0);
    }
    stage_b(4, 0, 0);
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        stage_a(g, 1, 1);
        stage_b(g, 1, 1);
    }
    wait_vmcnt<8>();
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();

    auto tile = [&](int t, int buf) {
        const char* A = smem + buf * A_BYTES + arow;
        const char* B = smem + buf * B_BYTES + brow;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            // ---- read section
            bf16x8_t fb[2];
            if (j == 0) {
                // only the first k-step's A fragments ahead of the barrier; the second k-step's are read inside the matrix section,
                // under this wave's own first seven MFMAs (the A rows of tile t are not restaged before phase (t, 2)): -0.7 %
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[i][0] = *reinterpret_cast<const bf16x8_t*>(A + i * 16 * RB + ((l4 ^ sw) << 4));
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fb[kk] = *reinterpret_cast<const bf16x8_t*>(B + j * 16 * RB + (((kk * 4 + l4) ^ sw) << 4));
            if (j >= 2) {
                stage_a(j - 2, t + 2, buf);
                stage_b(j - 2, t + 2, buf);
            } else {
                if (j == 0) stage_a(3, t + 1, buf ^ 1);
                stage_b(j + 3, t + 1, buf ^ 1);
            }
            // counted wait for what phase j+1 reads = the loads this wave issued after the last one that phase needs: phase (t+1, 0)
            // needs A3(t+1), issued first in phase (t, 0) -> B3 + 1 + 2 + 2 + 2 = 8 younger loads; B block g was issued seven phases
            // before the phase that reads it -> 12 or 13 younger loads
            const int jn = (j + 1) % NJ;
            if (jn == 0)
                wait_vmcnt<8>();
            else if (jn == 1 || jn == 4)
                wait_vmcnt<13>();
            else
                wait_vmcnt<12>();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- matrix section
            if (j == 0) {
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[i][1] = *reinterpret_cast<const bf16x8_t*>(A + i * 16 * RB + (((4 + l4) ^ sw) << 4));
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[i][j] = mfma_16x16x32(fa[i][kk], fb[kk], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int t = 0; t < nk; t += 2) {
        tile(t, 0);
        if (t + 1 < nk) tile(t + 1, 1);
    }
    }
    wait_vmcnt<0>();
    if (grp == 0) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
    gemm_epilogue16<MI, NJ>(p, acc, smem, (int)m0 + wm * 112, n0 + wn * 80, lane, wave, split);
}

// ---------------------------------------------------------------------------------------------
// k_gemm_p7x: k_gemm_p7 for the exact mode's split operands (GemmParams::split2).  The operand images are those of exact.py --
// A rows [a_hi | a_lo | a_hi] (3 * Cin channels per pixel), W rows [w_hi | w_hi | w_lo] in the usual K order over the 3 * Cin
// channels -- and the product is the same a_hi w_hi + a_lo w_hi + a_hi w_lo with fp32 accumulation.  k_gemm_p7 walks the 3K axis
// and stages every 64-wide K-tile of both images: per 64 channels of the ORIGINAL K three A tiles and three B tiles, of which one A
// tile (a_hi) and one B tile (w_hi) are staged twice.  Its K loop is bound by the staging (9 LDS-DMA instructions per wave and
// K-tile at 100-185 cycles each inside a read section against 1120 cycles of MFMA, MI355X_MICROARCH.md), so this kernel stages the
// FOUR distinct tiles of a 64-channel macro-tile once -- LDS map A[0] = a_hi, A[1] = a_lo, B[0] = w_hi, B[1] = w_lo, the planes
// read in place out of the 3-plane images -- and runs three K-steps on them:
//     s0: a_lo . w_hi  (A[1], B[0])      s1: a_hi . w_hi  (A[0], B[0])      s2: a_hi . w_lo  (A[0] fragments kept from s1, B[1])
// 18 DMA instructions per wave and macro-tile instead of 27, 58 fragment reads instead of 72, same 210 MFMAs.  A step keeps
// k_gemm_p7's five phases (B fragment column j per phase, A fragments read in phase 0, two barriers per phase, wave groups half a
// phase apart).  Staging schedule of macro-tile kc (piece = one A set / one B block, in per-wave issue order):
//     s0(kc): p0 A0(kc).1 | p1 A0(kc).2 | p2 A0(kc).3, A1(kc+1).0 | p3 A1(kc+1).1 | p4 A1(kc+1).2
//     s1(kc): pj B1(kc).j (j = 0..4), p4 also A1(kc+1).3
//     s2(kc): pj B0(kc+1).j (j = 0..4), p4 also A0(kc+1).0
// Every piece overwrites a region whose last read lies at least one full phase back (A rows are read in phase 0 only, B block j in
// phase j only; s2 reads no A rows), and lands at least two phases before it is read.  Counted waits (loads younger than the last
// one the next phase needs): s0 5,5,6,6,3; s1 -,-,-,-,5; s2 5,5,5,5,5.  One source only (exact.py materialises channel concats).
// <NJ = 4, G3>: the 224 x 256 instantiation for the GEGLU projection (wave tile 112 x 64 = 7 x 4 fragments, four phases per step, 16
// pieces per wave and macro-tile: s0 A0.1 A0.2 | A0.3 | A1'.0 A1'.1 | A1'.2, waits 5,5,6,3; s1 B1.j then A1'.3, wait -,-,-,4; s2 B0'.j
// then A0'.0, waits 4,4,4,4) with the GEGLU product formed in the epilogue and written as the split image (gemm_epilogue16<.., G3>).
// ---------------------------------------------------------------------------------------------
template <int NJ, bool G3>
__global__ void __launch_bounds__(512, 2) k_gemm_p7x(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MI = 7;
    constexpr int BM = 224, BN = NJ * 64, WN = NJ * 16, RB = 128;     // NJ = 5: 224 x 320; NJ = 4: 224 x 256 (GEGLU pairs of fragment columns)
    constexpr int A_BYTES = 256 * RB, B_BYTES = BN * RB;                 // LDS map: A[0] A[1] B[0] B[1] (hi, lo, hi, lo)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, grp = wave >> 2;
    int split, tn;
    long long tm;
    map_tile(p, BM, BN, split, tm, tn);
    const long long m0 = tm * BM;
    const int n0 = tn * BN;

    constexpr unsigned OOB = 0xF0000000u;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.w), 0, (int)((long long)p.N * p.K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x0), 0, (int)p.x0_bytes, 0x00020000);

    const int Cin = p.C0 / 3;                                            // channels of the ORIGINAL operand; a pixel holds [hi | lo | hi] = 3 * Cin
    const int HWo = p.Hout * p.Wout;
    const int upsh = p.up - 1;
    const int Hup = p.Hin << upsh, Wup = p.Win << upsh;
    const int lrow = lane >> 3, lch = lane & 7;
    const unsigned swz16 = (unsigned)((lch ^ (((wave & 1) << 2) | (lrow >> 1))) << 4);
    const int hb = p.ksize == 1 ? 1 : (p.tmode ? p.T : Hup), wb = (p.ksize == 1 || p.tmode) ? 1 : Wup;
    const int wmul = p.tmode ? HWo : p.Win;
    int a_base[4], a_hw[4];                                            // a_hw = (ih0 + 0x4000) << 16 | (iw0 + 0x4000)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const int r = (s4 >> 1) * 128 + (wave + 8 * (s4 & 1)) * 8 + lrow;
        const long long m = m0 + r;
        const bool ok = r < BM && m < p.M;
        const int mm = ok ? (int)m : 0;
        int ih0 = 0, iw0 = 0;
        if (p.ksize == 1) {
            a_base[s4] = mm;
        } else if (p.tmode) {
            const int t = (mm / HWo) % p.T;
            a_base[s4] = mm - t * HWo;
            ih0 = t - 1;
        } else {
            const int b = mm / HWo, rem = mm - b * HWo;
            const int oh = rem / p.Wout, ow = rem - oh * p.Wout;
            a_base[s4] = b * p.Hin * p.Win;
            ih0 = oh * p.stride - p.pad;
            iw0 = ow * p.stride - p.pad;
        }
        if (!ok) ih0 = -0x4000;
        a_hw[s4] = ((ih0 + 0x4000) << 16) | (iw0 + 0x4000);
    }
    const int b_r0 = (wave >> 1) * WN + (wave & 1) * 8;
    const int b_n = n0 + b_r0 + lrow;
    const unsigned b_off0 = (unsigned)((long long)b_n * p.K * 2) + swz16;
    const unsigned b_gstep = (unsigned)p.K * 32u;                            // 16 weight rows
    const unsigned b_lo = (unsigned)(p.K / 3) * 4u;                          // w_lo = third plane of the row: 2 * (K / 3) elements further
    const int kchunk = p.ksize == 1 ? Cin : 64;                              // K order of the ORIGINAL axis (chunk-major for convolutions)
    const int nk_all = p.K / 192;                                            // macro-tiles: 64 channels of the original K
    const int ks_begin = p.ksplit > 1 ? (int)((long long)nk_all * split / p.ksplit) : 0;
    const int ks_end = p.ksplit > 1 ? (int)((long long)nk_all * (split + 1) / p.ksplit) : nk_all;
    const int nk = ks_end - ks_begin;
    // one K cursor per A plane: the hi plane runs one macro-tile behind the lo plane (see the schedule)
    KCursor cur0, cur1;
    cur0.init(ks_begin * 64, p.taps, kchunk);
    cur1 = cur0;
    int u0 = 0, u1 = 0;                                                       // macro-tile each cursor points at
    int k0h = 0, k0w = 0, k0c = 0, k1h = 0, k1w = 0, k1c = 0;
    bool live0 = false, live1 = false;
    auto stage_a = [&](auto PL, int s4) {                                     // PL: 0 = hi plane -> A[0], 1 = lo plane -> A[1]
        constexpr int pl = decltype(PL)::value;
        KCursor& cur = pl ? cur1 : cur0;
        int& kh = pl ? k1h : k0h;
        int& kw = pl ? k1w : k0w;
        int& cc = pl ? k1c : k0c;
        bool& live = pl ? live1 : live0;
        int& u = pl ? u1 : u0;
        if (s4 == 0) {
            live = u < nk;
            ++u;
            cc = cur.c0();
            const int t3 = cur.tap / 3;
            kh = p.tmode ? cur.tap : t3;
            kw = p.tmode ? 0 : cur.tap - t3 * 3;
            cur.advance(64, p.taps, kchunk);
        }
        char* dst = smem + pl * A_BYTES + ((s4 >> 1) * 128 + (wave + 8 * (s4 & 1)) * 8) * RB;
        const int ih = (a_hw[s4] >> 16) - 0x4000 + kh, iw = (a_hw[s4] & 0xFFFF) - 0x4000 + kw;
        const bool ok = live && (unsigned)ih < (unsigned)hb && (unsigned)iw < (unsigned)wb;
        const int pix = a_base[s4] + (ih >> upsh) * wmul + (iw >> upsh);
        const unsigned off = ok ? (unsigned)(pix * p.C0 + pl * Cin + cc) * 2u + swz16 : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x0, (lds_ptr_t)dst, 16, (int)off, 0, 0, 0);
    };
    auto stage_b = [&](auto PL, int g, int u) {                               // PL: 0 = w_hi -> B[0], 1 = w_lo -> B[1]; u = macro-tile
        constexpr int pl = decltype(PL)::value;
        const bool live = u < nk;
        char* dst = smem + 2 * A_BYTES + pl * B_BYTES + (b_r0 + g * 16) * RB;
        const unsigned off = (live && b_n + g * 16 < p.N) ? b_off0 + (unsigned)g * b_gstep + (pl ? b_lo : 0u) + (unsigned)(ks_begin + u) * 128u : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)dst, 16, (int)off, 0, 0, 0);
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int sw = (l15 >> 1) & 7;
    const int arow = (wm * 112 + l15) * RB, brow = 2 * A_BYTES + (wn * WN + l15) * RB;

    // prologue: what the steady state would have issued before s0(0), in its issue order
#pragma unroll
    for (int g = 0; g < 4; ++g) stage_a(P1{}, g);                             // A1(0)
#pragma unroll
    for (int g = 0; g < NJ; ++g) stage_b(P0{}, g, 0);                         // B0(0)
    stage_a(P0{}, 0);                                                         // A0(0).0
    wait_vmcnt<NJ>();
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();

    bf16x8_t fa[MI][2];
    auto step = [&](auto SK, int kc) {                                        // SK: 0 a_lo.w_hi, 1 a_hi.w_hi, 2 a_hi.w_lo of macro-tile kc
        constexpr int sk = decltype(SK)::value;
        const char* A = smem + (sk == 0 ? A_BYTES : 0) + arow;
        const char* B = smem + (sk == 2 ? B_BYTES : 0) + brow;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            // ---- read section
            bf16x8_t fb[2];
            if (j == 0 && sk != 2) {
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[i][0] = *reinterpret_cast<const bf16x8_t*>(A + i * 16 * RB + ((l4 ^ sw) << 4));
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fb[kk] = *reinterpret_cast<const bf16x8_t*>(B + j * 16 * RB + (((kk * 4 + l4) ^ sw) << 4));
            if (sk == 0) {
                if constexpr (NJ == 5) {
                    if (j <= 2) stage_a(P0{}, j + 1);                         // A0(kc).1..3
                    if (j >= 2) stage_a(P1{}, j - 2);                         // A1(kc+1).0..2
                    if (j <= 1) wait_vmcnt<5>();
                    else if (j <= 3) wait_vmcnt<6>();
                    else wait_vmcnt<3>();
                } else {                                                      // four phases: A0.1 A0.2 | A0.3 | A1'.0 A1'.1 | A1'.2; waits 5, 5, 6, 3
                    if (j == 0) {
                        stage_a(P0{}, 1);
                        stage_a(P0{}, 2);
                    } else if (j == 1) {
                        stage_a(P0{}, 3);
                    } else if (j == 2) {
                        stage_a(P1{}, 0);
                        stage_a(P1{}, 1);
                    } else {
                        stage_a(P1{}, 2);
                    }
                    if (j <= 1) wait_vmcnt<5>();
                    else if (j == 2) wait_vmcnt<6>();
                    else wait_vmcnt<3>();
                }
            } else if (sk == 1) {
                stage_b(P1{}, j, kc);                                         // B1(kc).j
                if (j == NJ - 1) {
                    stage_a(P1{}, 3);                                         // A1(kc+1).3
                    wait_vmcnt<NJ>();
                }
            } else {
                stage_b(P0{}, j, kc + 1);                                     // B0(kc+1).j
                if (j == NJ - 1) stage_a(P0{}, 0);                            // A0(kc+1).0
                wait_vmcnt<NJ>();
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- matrix section
            if (j == 0 && sk != 2) {
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[i][1] = *reinterpret_cast<const bf16x8_t*>(A + i * 16 * RB + (((4 + l4) ^ sw) << 4));
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[i][j] = mfma_16x16x32(fa[i][kk], fb[kk], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    for (int kc = 0; kc < nk; ++kc) {
        step(S0{}, kc);
        step(S1{}, kc);
        step(S2{}, kc);
    }
    wait_vmcnt<0>();
    if (grp == 0) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
    gemm_epilogue16<MI, NJ, G3>(p, acc, smem, (int)m0 + wm * 112, n0 + wn * WN, lane, wave, split);
}

// ---------------------------------------------------------------------------------------------
// k_gemm_phx: the 256 x 320 phased tile (k_gemm_ph<5>: 4 x 2 waves, wave tile 64 x 160 of v_mfma_f32_32x32x16, B blocks of 32 rows)
// on the exact mode's split operands with k_gemm_p7x's staging: the four distinct tiles of a 64-channel macro-tile (a_hi, a_lo, w_hi,
// w_lo) staged once, three K-steps on them, the same schedule and counted waits (the piece counts per wave and tile are the same: four
// A sets, five B blocks).  For the shapes whose M fills whole rounds of 256-row tiles -- the 36x64 and 18x32 levels of the SVD window.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512, 2) k_gemm_phx(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NJ = 5;
    constexpr int BM = 256, BN = NJ * 64, RB = 128;
    constexpr int A_BYTES = 256 * RB, B_BYTES = BN * RB;                 // LDS map: A[0] A[1] B[0] B[1] (hi, lo, hi, lo)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, grp = wave >> 2;
    int split, tn;
    long long tm;
    map_tile(p, BM, BN, split, tm, tn);
    const long long m0 = tm * BM;
    const int n0 = tn * BN;

    constexpr unsigned OOB = 0xF0000000u;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.w), 0, (int)((long long)p.N * p.K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x0), 0, (int)p.x0_bytes, 0x00020000);

    const int Cin = p.C0 / 3;                                            // channels of the ORIGINAL operand; a pixel holds [hi | lo | hi] = 3 * Cin
    const int HWo = p.Hout * p.Wout;
    const int upsh = p.up - 1;
    const int Hup = p.Hin << upsh, Wup = p.Win << upsh;
    const int lrow = lane >> 3, lch = lane & 7;
    const unsigned swz16 = (unsigned)((lch ^ (((wave & 1) << 2) | (lrow >> 1))) << 4);
    const int hb = p.ksize == 1 ? 1 : (p.tmode ? p.T : Hup), wb = (p.ksize == 1 || p.tmode) ? 1 : Wup;
    const int wmul = p.tmode ? HWo : p.Win;
    int a_base[4], a_hw[4];                                            // a_hw = (ih0 + 0x4000) << 16 | (iw0 + 0x4000)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const int r = (s4 >> 1) * 128 + (wave + 8 * (s4 & 1)) * 8 + lrow;
        const long long m = m0 + r;
        const bool ok = m < p.M;
        const int mm = ok ? (int)m : 0;
        int ih0 = 0, iw0 = 0;
        if (p.ksize == 1) {
            a_base[s4] = mm;
        } else if (p.tmode) {
            const int t = (mm / HWo) % p.T;
            a_base[s4] = mm - t * HWo;
            ih0 = t - 1;
        } else {
            const int b = mm / HWo, rem = mm - b * HWo;
            const int oh = rem / p.Wout, ow = rem - oh * p.Wout;
            a_base[s4] = b * p.Hin * p.Win;
            ih0 = oh * p.stride - p.pad;
            iw0 = ow * p.stride - p.pad;
        }
        if (!ok) ih0 = -0x4000;
        a_hw[s4] = ((ih0 + 0x4000) << 16) | (iw0 + 0x4000);
    }
    const int b_r0 = (wave >> 2) * (NJ * 32) + (wave & 3) * 8;           // first row of this wave's piece inside B block 0 (k_gemm_ph)
    const int b_n = n0 + b_r0 + lrow;
    const unsigned b_off0 = (unsigned)((long long)b_n * p.K * 2) + swz16;
    const unsigned b_gstep = (unsigned)p.K * 64u;                            // 32 weight rows
    const unsigned b_lo = (unsigned)(p.K / 3) * 4u;                          // w_lo = third plane of the row: 2 * (K / 3) elements further
    const int kchunk = p.ksize == 1 ? Cin : 64;                              // K order of the ORIGINAL axis (chunk-major for convolutions)
    const int nk_all = p.K / 192;                                            // macro-tiles: 64 channels of the original K
    const int ks_begin = p.ksplit > 1 ? (int)((long long)nk_all * split / p.ksplit) : 0;
    const int ks_end = p.ksplit > 1 ? (int)((long long)nk_all * (split + 1) / p.ksplit) : nk_all;
    const int nk = ks_end - ks_begin;
    // one K cursor per A plane: the hi plane runs one macro-tile behind the lo plane (see the schedule)
    KCursor cur0, cur1;
    cur0.init(ks_begin * 64, p.taps, kchunk);
    cur1 = cur0;
    int u0 = 0, u1 = 0;                                                       // macro-tile each cursor points at
    int k0h = 0, k0w = 0, k0c = 0, k1h = 0, k1w = 0, k1c = 0;
    bool live0 = false, live1 = false;
    auto stage_a = [&](auto PL, int s4) {                                     // PL: 0 = hi plane -> A[0], 1 = lo plane -> A[1]
        constexpr int pl = decltype(PL)::value;
        KCursor& cur = pl ? cur1 : cur0;
        int& kh = pl ? k1h : k0h;
        int& kw = pl ? k1w : k0w;
        int& cc = pl ? k1c : k0c;
        bool& live = pl ? live1 : live0;
        int& u = pl ? u1 : u0;
        if (s4 == 0) {
            live = u < nk;
            ++u;
            cc = cur.c0();
            const int t3 = cur.tap / 3;
            kh = p.tmode ? cur.tap : t3;
            kw = p.tmode ? 0 : cur.tap - t3 * 3;
            cur.advance(64, p.taps, kchunk);
        }
        char* dst = smem + pl * A_BYTES + ((s4 >> 1) * 128 + (wave + 8 * (s4 & 1)) * 8) * RB;
        const int ih = (a_hw[s4] >> 16) - 0x4000 + kh, iw = (a_hw[s4] & 0xFFFF) - 0x4000 + kw;
        const bool ok = live && (unsigned)ih < (unsigned)hb && (unsigned)iw < (unsigned)wb;
        const int pix = a_base[s4] + (ih >> upsh) * wmul + (iw >> upsh);
        const unsigned off = ok ? (unsigned)(pix * p.C0 + pl * Cin + cc) * 2u + swz16 : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x0, (lds_ptr_t)dst, 16, (int)off, 0, 0, 0);
    };
    auto stage_b = [&](auto PL, int g, int u) {                               // PL: 0 = w_hi -> B[0], 1 = w_lo -> B[1]; u = macro-tile
        constexpr int pl = decltype(PL)::value;
        const bool live = u < nk;
        char* dst = smem + 2 * A_BYTES + pl * B_BYTES + (b_r0 + g * 32) * RB;
        const unsigned off = (live && b_n + g * 32 < p.N) ? b_off0 + (unsigned)g * b_gstep + (pl ? b_lo : 0u) + (unsigned)(ks_begin + u) * 128u : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)dst, 16, (int)off, 0, 0, 0);
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int l31 = lane & 31, hi = lane >> 5;
    const int sw = lds_swz<64>(l31);
    const int arow = (wm * 64 + l31) * RB, brow = 2 * A_BYTES + (wn * (NJ * 32) + l31) * RB;

    // prologue: what the steady state would have issued before s0(0), in its issue order
#pragma unroll
    for (int g = 0; g < 4; ++g) stage_a(P1{}, g);                             // A1(0)
#pragma unroll
    for (int g = 0; g < NJ; ++g) stage_b(P0{}, g, 0);                         // B0(0)
    stage_a(P0{}, 0);                                                         // A0(0).0
    wait_vmcnt<NJ>();
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();

    bf16x8_t fa[2][4];
    auto step = [&](auto SK, int kc) {                                        // SK: 0 a_lo.w_hi, 1 a_hi.w_hi, 2 a_hi.w_lo of macro-tile kc
        constexpr int sk = decltype(SK)::value;
        const char* A = smem + (sk == 0 ? A_BYTES : 0) + arow;
        const char* B = smem + (sk == 2 ? B_BYTES : 0) + brow;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            // ---- read section
            bf16x8_t fb[4];
            if (j == 0 && sk != 2) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int i = 0; i < 2; ++i) fa[i][s4] = *reinterpret_cast<const bf16x8_t*>(A + i * 32 * RB + (((s4 * 2 + hi) ^ sw) << 4));
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) fb[s4] = *reinterpret_cast<const bf16x8_t*>(B + j * 32 * RB + (((s4 * 2 + hi) ^ sw) << 4));
            if (sk == 0) {
                if constexpr (NJ == 5) {
                    if (j <= 2) stage_a(P0{}, j + 1);                         // A0(kc).1..3
                    if (j >= 2) stage_a(P1{}, j - 2);                         // A1(kc+1).0..2
                    if (j <= 1) wait_vmcnt<5>();
                    else if (j <= 3) wait_vmcnt<6>();
                    else wait_vmcnt<3>();
                } else {                                                      // four phases: A0.1 A0.2 | A0.3 | A1'.0 A1'.1 | A1'.2; waits 5, 5, 6, 3
                    if (j == 0) {
                        stage_a(P0{}, 1);
                        stage_a(P0{}, 2);
                    } else if (j == 1) {
                        stage_a(P0{}, 3);
                    } else if (j == 2) {
                        stage_a(P1{}, 0);
                        stage_a(P1{}, 1);
                    } else {
                        stage_a(P1{}, 2);
                    }
                    if (j <= 1) wait_vmcnt<5>();
                    else if (j == 2) wait_vmcnt<6>();
                    else wait_vmcnt<3>();
                }
            } else if (sk == 1) {
                stage_b(P1{}, j, kc);                                         // B1(kc).j
                if (j == NJ - 1) {
                    stage_a(P1{}, 3);                                         // A1(kc+1).3
                    wait_vmcnt<NJ>();
                }
            } else {
                stage_b(P0{}, j, kc + 1);                                     // B0(kc+1).j
                if (j == NJ - 1) stage_a(P0{}, 0);                            // A0(kc+1).0
                wait_vmcnt<NJ>();
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- matrix section
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i][j] = mfma_32x32x16(fa[i][s4], fb[s4], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    for (int kc = 0; kc < nk; ++kc) {
        step(S0{}, kc);
        step(S1{}, kc);
        step(S2{}, kc);
    }
    wait_vmcnt<0>();
    if (grp == 0) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
    gemm_epilogue<NJ, 2, true, false>(p, acc, smem, m0 + wm * 64, n0 + wn * (NJ * 32), lane, wave, split);
}

// Split-K finish: sum the fp32 partials in split order (deterministic), then the same epilogue as above.
__global__ void __launch_bounds__(256) k_splitk_finish(GemmParams p) {
    const long long i8 = (long long)blockIdx.x * 256 + threadIdx.x;
    const int n8 = p.N / 8;
    if (i8 >= p.M * n8) return;
    const int m = (int)(i8 / n8);                                 // rows fit 31 bits
    const int n = (int)(i8 - (long long)m * n8) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    const float* wp = p.ws + (long long)m * p.N + n;
    const long long sstride = p.M * p.N;
    for (int s = 0; s < p.ksplit; ++s) {                           // ascending split order: deterministic
        const f32x4 a = *reinterpret_cast<const f32x4*>(wp), b = *reinterpret_cast<const f32x4*>(wp + 4);
        wp += sstride;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] += a[e];
            v[4 + e] += b[e];
        }
    }
    if (p.bias) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n), b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] += b0[e];
            v[4 + e] += b1[e];
        }
    }
    if (p.tap && p.tap_early && n < p.tap_cols) {
        f16x8 t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = (f16)v[e];
        *reinterpret_cast<f16x8*>(p.tap + (long long)m * p.tap_ld + n) = t;
    }
    if (p.rowvec) {
        const float* rvp = p.rowvec + (long long)(m / p.rows_per_sample) * p.rv_stride + n;
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(rvp), r1 = *reinterpret_cast<const f32x4*>(rvp + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] += r0[e];
            v[4 + e] += r1[e];
        }
    }
    if (p.act == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
    }
    if (p.rowadd) {
        const float ra = p.rowadd[m];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += ra;
    }
    if (p.tap && !p.tap_early && n < p.tap_cols) {
        f16x8 t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = (f16)v[e];
        *reinterpret_cast<f16x8*>(p.tap + tap_row(p, m) * p.tap_ld + n) = t;
    }
    if (p.tap2 && n >= p.tap_cols && n < 2 * p.tap_cols) {
        f16x8 t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = (f16)v[e];
        *reinterpret_cast<f16x8*>(p.tap2 + tap_row(p, m) * p.tap_ld + (n - p.tap_cols)) = t;
    }
    if (p.residual && p.res_f32) {
        const float* rp = reinterpret_cast<const float*>(p.residual) + (long long)m * p.ldr + n;
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] += r0[e];
            v[4 + e] += r1[e];
        }
    } else if (p.residual) {
        const bf16x8_t rr = *reinterpret_cast<const bf16x8_t*>(p.residual + (long long)m * p.ldr + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bf16_to_f32((bf16_t)rr[e]);
    }
    if (p.blend) {                                              // as in epilogue_rows
        const float* bp = p.blend + (long long)m * p.ldo + n;
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(bp), s1 = *reinterpret_cast<const f32x4*>(bp + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = p.blend_a * s0[e] + p.blend_b * v[e];
            v[4 + e] = p.blend_a * s1[e] + p.blend_b * v[4 + e];
        }
    }
    if (p.plane_hi && n >= p.plane_col0) {                      // as in epilogue_rows
        f16x8 h8, l8;
        split_cell8(v, h8, l8);
        const long long po = (long long)m * p.plane_ld + (n - p.plane_col0);
        *reinterpret_cast<f16x8*>(p.plane_hi + po) = h8;
        *reinterpret_cast<f16x8*>(p.plane_lo + po) = l8;
        return;
    }
    if (p.out) {
        bf16x8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (short)f32_to_bf16(v[e]);
        *reinterpret_cast<bf16x8_t*>(p.out + (long long)m * p.ldo + n) = o;
    }
    if (p.out_f32) {
        f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
        *reinterpret_cast<f32x4*>(p.out_f32 + (long long)m * p.ldo + n) = a;
        *reinterpret_cast<f32x4*>(p.out_f32 + (long long)m * p.ldo + n + 4) = b;
    }
    if (p.out_split3) {                                        // as in epilogue_rows
        f16x8 h8, l8;
        split_cell8(v, h8, l8);
        f16* o3 = p.out_split3 + (long long)m * 3 * p.ldo + n;
        *reinterpret_cast<f16x8*>(o3) = h8;
        *reinterpret_cast<f16x8*>(o3 + p.ldo) = l8;
        if (VS_THIRD_PLANE(p.ldo)) *reinterpret_cast<f16x8*>(o3 + 2 * p.ldo) = h8;
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

// Input conv (Cin = 3/4/8, Cout multiple of 8): the [3][3][Cin][Cout] fp32 weights live in LDS for the block's lifetime;
// thread t owns 8 consecutive output channels (t % (Cout/8)) of a QUAD of 4 horizontally adjacent pixels: per (kh, c) it
// loads the 6 inputs the quad's three kw taps touch (same address for all threads of a quad: broadcast) and reuses every
// weight vector for the 4 pixels -- 96 FMAs per 6 LDS reads, VALU-bound.  The bf16 result leaves as 16-byte stores.
#define CONV_IN_QUADS 32
__global__ void __launch_bounds__(256) k_conv_in(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                 int B, int H, int W, int Cin, int Cout, bf16_t* __restrict__ out, float* __restrict__ out32) {
    extern __shared__ __attribute__((aligned(16))) float wl[];           // [9*Cin][Cout]
    const int K = 9 * Cin;
    for (int i = threadIdx.x; i < K * Cout / 4; i += 256) reinterpret_cast<f32x4*>(wl)[i] = reinterpret_cast<const f32x4*>(w)[i];
    __syncthreads();
    const int c8n = Cout / 8;
    const int slots = 256 / c8n;                                         // quads in flight per pass
    const int slot = threadIdx.x / c8n, co = (threadIdx.x - slot * c8n) * 8;
    if (slot >= slots) return;
    const int W4 = W / 4;
    const long long nquad = (long long)B * H * W4;
    const long long q0 = (long long)blockIdx.x * CONV_IN_QUADS;
    float bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[j] = bias ? bias[co + j] : 0.f;
    for (int qi = slot; qi < CONV_IN_QUADS; qi += slots) {
        const long long quad = q0 + qi;
        if (quad >= nquad) break;
        const int ow0 = (int)(quad % W4) * 4, oh = (int)((quad / W4) % H), b = (int)(quad / ((long long)W4 * H));
        float acc[4][8];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[p][j] = bv[j];
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh + kh - 1;
            if (ih < 0 || ih >= H) continue;
            const float* xrow = x + ((long long)b * H + ih) * W * Cin;
            for (int c = 0; c < Cin; ++c) {
                float xv[6];
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    const int iw = ow0 - 1 + t;
                    xv[t] = (iw >= 0 && iw < W) ? xrow[(long long)iw * Cin + c] : 0.f;
                }
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float* wp = wl + ((kh * 3 + kw) * Cin + c) * Cout + co;
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp), w1 = *reinterpret_cast<const f32x4*>(wp + 4);
#pragma unroll
                    for (int p = 0; p < 4; ++p)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            acc[p][j] = fmaf(xv[kw + p], w0[j], acc[p][j]);
                            acc[p][4 + j] = fmaf(xv[kw + p], w1[j], acc[p][4 + j]);
                        }
                }
            }
        }
        if (out32) {                                           // exact mode: the fp32 accumulators as they are
            float* op32 = out32 + (((long long)b * H + oh) * W + ow0) * Cout + co;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                *reinterpret_cast<f32x4*>(op32 + (long long)p * Cout) = f32x4{acc[p][0], acc[p][1], acc[p][2], acc[p][3]};
                *reinterpret_cast<f32x4*>(op32 + (long long)p * Cout + 4) = f32x4{acc[p][4], acc[p][5], acc[p][6], acc[p][7]};
            }
            continue;
        }
        bf16_t* op = out + (((long long)b * H + oh) * W + ow0) * Cout + co;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            u32x4 o = {pack2_bf16(acc[p][0], acc[p][1]), pack2_bf16(acc[p][2], acc[p][3]), pack2_bf16(acc[p][4], acc[p][5]),
                       pack2_bf16(acc[p][6], acc[p][7])};
            *reinterpret_cast<u32x4*>(op + (long long)p * Cout) = o;
        }
    }
}

// Output conv (Cout = 4, Cin multiple of 8): one wave per output pixel; lanes split the 9*Cin reduction in
// 16-byte pieces, weights [Cout][3][3][Cin] bf16, wave-shuffle reduction, fp32 NCHW result.
__global__ void __launch_bounds__(256) k_conv_out4(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, const float* __restrict__ bias,
                                                   int B, int H, int W, int Cin, float* __restrict__ out_nchw) {
    const int lane = threadIdx.x & 63;
    const long long pix = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= (long long)B * H * W) return;
    const int ow = (int)(pix % W), oh = (int)((pix / W) % H), b = (int)(pix / ((long long)W * H));
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int c8 = Cin / 8;
    for (int t = lane; t < 9 * c8; t += 64) {
        const int tap = t / c8, c = (t % c8) * 8;
        const int ih = oh + tap / 3 - 1, iw = ow + tap % 3 - 1;
        if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
        const bf16x8_t xv = *reinterpret_cast<const bf16x8_t*>(x + (((long long)b * H + ih) * W + iw) * Cin + c);
#pragma unroll
        for (int co = 0; co < 4; ++co) {
            const bf16x8_t wv = *reinterpret_cast<const bf16x8_t*>(w + ((long long)co * 9 + tap) * Cin + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[co] = fmaf(bf16_to_f32((bf16_t)xv[e]), bf16_to_f32((bf16_t)wv[e]), acc[co]);
        }
    }
#pragma unroll
    for (int co = 0; co < 4; ++co) acc[co] = wave_sum_f32(acc[co]);
    if (lane == 0)
        for (int co = 0; co < 4; ++co) out_nchw[(((long long)b * 4 + co) * H + oh) * W + ow] = acc[co] + (bias ? bias[co] : 0.f);
}

// The same with the weights held in registers: a wave keeps its lanes' 16-byte pieces of all four filters (NP pieces per lane,
// 9 * Cin / 8 <= 64 NP) and walks a strided list of output pixels, so the 23 KB of weights are read once per wave instead of once
// per pixel (k_conv_out4 at 28 x 64 x 64 x 320: 175 -> 106 us for a 73 MB input); products on v_dot2c_f32_{f16,bf16} (v_fma_mix_f32: 122 us).
// What is left is L2 traffic: neighbouring pixels run on different CUs, so the 9 taps of a pixel are 9 L2 reads of its 640-byte rows;
// an LDS-tiled version (8 x 8 pixels + halo per block) would cut that to ~1.6x.
template <int NP>
__global__ void __launch_bounds__(256) k_conv_out4_ws(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, const float* __restrict__ bias,
                                                      int B, int H, int W, int Cin, float* __restrict__ out_nchw) {
    const int lane = threadIdx.x & 63;
    const long long npix = (long long)B * H * W;
    const int c8 = Cin / 8, np = 9 * c8;
    u32x4 wreg[NP][4];
    int dh[NP], dw[NP], cofs[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int t = min(lane + 64 * k, np - 1), tap = t / c8;
        dh[k] = tap / 3 - 1;
        dw[k] = tap % 3 - 1;
        cofs[k] = (t - tap * c8) * 8;
        if (lane + 64 * k >= np) dh[k] = 1 << 20;              // no such piece: always out of range
#pragma unroll
        for (int co = 0; co < 4; ++co) wreg[k][co] = *reinterpret_cast<const u32x4*>(w + ((long long)co * 9 + tap) * Cin + cofs[k]);
    }
    const float b_lane = (bias && lane < 4) ? bias[lane] : 0.f;
    const long long stride = (long long)gridDim.x * 4;
    // two pixels per iteration: both pixels' pieces are requested before either's products (a wave has one SIMD partner: 170+ registers)
    u32x4 xv[2][NP];
    bool ok[2][NP];
    for (long long pix0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); pix0 < npix; pix0 += 2 * stride) {
        int ow[2], oh[2], bb[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long pc = pix0 + u * stride < npix ? pix0 + u * stride : pix0;
            ow[u] = (int)(pc % W);
            oh[u] = (int)((pc / W) % H);
            bb[u] = (int)(pc / ((long long)W * H));
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int ih = oh[u] + dh[k], iw = ow[u] + dw[k];
                ok[u][k] = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                const long long off = ok[u][k] ? (((long long)bb[u] * H + ih) * W + iw) * Cin + cofs[k] : 0;
                xv[u][k] = *reinterpret_cast<const u32x4*>(x + off);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                if (!ok[u][k]) continue;
#pragma unroll
                for (int co = 0; co < 4; ++co)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[co] = dot2_acc(xv[u][k][e], wreg[k][co][e], acc[co]);
            }
#pragma unroll
            for (int co = 0; co < 4; ++co) acc[co] = wave_sum_f32(acc[co]);
            const float v = lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : acc[3];
            if (lane < 4 && (u == 0 || pix0 + stride < npix)) out_nchw[(((long long)bb[u] * 4 + lane) * H + oh[u]) * W + ow[u]] = v + b_lane;
        }
    }
}

// ---- live HIP-event timing of this kernel family (bench.py roofline) -------------------------------------
// When enabled, every k_gemm_conv launch is bracketed by two events recorded on the launch stream; collect()
// synchronises, sums the elapsed times and the algorithmic FLOPs (2*M*N*K per launch).
#include <vector>
#include <stdlib.h>
// Kernel selection is a fixed function of the problem (the decision table in launch_gemm).  ONE override exists, for the per-kernel
// tests and same-box A/B runs: VIDSEG_GEMM="key=value,key=value", read once per process.  Keys (default):
//   big (1)   0 never / 1 where the table picks it / 2 whenever legal -- the 8-wave phased tiles (k_gemm_ph 256x320 / 256x256, k_gemm_p7 224x320)
//   p7 (1)    0 / 1 / 2 likewise for the 224-row tile among the big ones;  p7x (1): 0 keeps k_gemm_p7 on split operands;
//             xsmall (1): 0 = split operands obey the 16-bit thresholds of the big tile (fill >= 0.7, K / S >= 1440);
//             phx (1): 0 keeps k_gemm_ph<5> on split operands
//   p7ph (5)  phases per K-tile of k_gemm_p7 (5 or 3);  ph (1): 0 = the unphased k_gemm_tile for the big shapes
//   mid (1)   0 / 1 / 2 the 128x320 tile;  dma (1): 0 = k_gemm_conv<128,128>, 3 = the 3-stage 128x128 everywhere;  tile (0): 128 forbids the narrow 256x64 tile
//   ws (1)    0 / 1 / 2 the weight-stationary streaming kernel;  convout (1): 0 = the plain 4-channel output conv
//   split (1) 0 = no split-K;  panel (1): 0 = row-major tile order
//   ext (1)   0 = hipEventRecord pairs instead of dispatch-packet timestamps (profiling);  fence (0): 1 = system-scope fence at the events
//   shapes (0) 1 = one GEMMSHAPE line per profiled launch on stderr (tools/shape_summary.py)
//   gg / gn (0) > 0 = a fixed panel width of the tile order for the GEGLU tile / every other tile (tools/panel_sweep.py)
struct GemmKnobs {
    int big = 1, p7 = 1, p7x = 1, phx = 1, xsmall = 1, p7ph = 5, ph = 1, mid = 1, dma = 1, tile = 0, ws = 1, convout = 1, split = 1, panel = 1, ext = 1, fence = 0,
        shapes = 0, gg = 0, gn = 0;
};
static const GemmKnobs& knobs() {
    static const GemmKnobs k = [] {
        GemmKnobs g;
        const char* e = getenv("VIDSEG_GEMM");
        if (!e) return g;
        struct { const char* name; int* v; } tab[] = {{"big", &g.big}, {"p7", &g.p7}, {"p7x", &g.p7x}, {"phx", &g.phx}, {"xsmall", &g.xsmall}, {"p7ph", &g.p7ph}, {"ph", &g.ph}, {"mid", &g.mid},
                                                      {"dma", &g.dma}, {"tile", &g.tile}, {"ws", &g.ws}, {"convout", &g.convout}, {"split", &g.split},
                                                      {"panel", &g.panel}, {"ext", &g.ext}, {"fence", &g.fence}, {"shapes", &g.shapes}, {"gg", &g.gg}, {"gn", &g.gn}};
        while (*e) {
            const char* eq = strchr(e, '=');
            if (!eq) break;
            bool known = false;
            for (auto& t : tab)
                if ((size_t)(eq - e) == strlen(t.name) && !strncmp(e, t.name, eq - e)) {
                    *t.v = atoi(eq + 1);
                    known = true;
                }
            if (!known) fprintf(stderr, "vidseg: VIDSEG_GEMM: unknown key in '%s'\n", e);
            const char* c = strchr(eq, ',');
            if (!c) break;
            e = c + 1;
        }
        return g;
    }();
    return k;
}
struct GemmProf {
    bool on = false;
    std::thread::id owner;           // the thread whose launches are being recorded (valid while on): its t_prof points here
    std::vector<hipEvent_t> ev;      // pairs
    size_t used = 0;
    double flops = 0.0;
    long long launches = 0;
    struct Shape { long long M; int N, K, ksize, up, stride, act, ksplit, kind; double alg_bytes; };
    std::vector<Shape> shapes;      // one per event pair
    double kind_stats[24] = {0};    // per kernel kind: ms, flops, launches (filled by _end)
    double kind_bytes[8] = {0};
};
// A profiler is a caller-owned handle (vidseg_gemm_profiler_create); between _begin(h) and _end(h) the GEMM launches of the CALLING
// THREAD are timed into it.  No profiler state lives in the library beyond the per-thread pointer to the handle that is recording.
static GemmProf g_off;                                          // the "not recording" profiler (on = false)
static thread_local GemmProf* t_prof = &g_off;
#define g_prof (*t_prof)

static inline hipEvent_t prof_event() {
    if (g_prof.used == g_prof.ev.size()) {
        hipEvent_t e;
        // no system-scope release at the record: the default flag flushes the L2 to make the kernels' results host-visible at
        // every event (measured: the 1200 records of a window cost 4-5 % of the step); timing needs no such fence
        static const unsigned flags = knobs().fence ? hipEventDefault : hipEventDisableSystemFence;
        (void)hipEventCreateWithFlags(&e, flags);
        g_prof.ev.push_back(e);
    }
    return g_prof.ev[g_prof.used++];
}

// ---------------------------------------------------------------------------------------------
// k_gemm_ws: weight-stationary streaming GEMM for the short-K linears (K = 320 at the 64x64 level, K = 640 at 32x32), the
// shapes where every tiled kernel above spends more time in its per-tile prologue / epilogue than in its K loop (114688 x 320 x
// 320: 54 us against an 18 us HBM floor).  One persistent 8-wave block per CU keeps a W panel of BN = 16 NJ columns x all of K
// in LDS (100 KB) for its whole life; every wave then streams its own 32-row tiles of A straight from global memory into MFMA
// operand registers (a lane's 16 bytes = 8 consecutive k of one row; a ring of ten k-steps, refilled as it is consumed, runs
// across tile boundaries) -- no A staging, no barrier after the panel load, waves never wait for each other.  The product is
// formed transposed (W fragment as the MFMA A operand), so a lane holds four consecutive output columns of one row and the
// epilogue stages 16-byte pieces through a per-wave fp32 buffer into the same epilogue_rows as every other kernel.
// Blocks on one XCD (blockIdx & 7) take the same row ranges for the different panels, so A is re-read from that XCD's L2.
// ---------------------------------------------------------------------------------------------
// EPI 0: the shared epilogue_rows (every option); 1 / 2: bias (+ residual) and a 16-bit result only, straight-line: a lane's
// three 8-column cells of an 8-row pass sit at fixed columns, so their bias lives in registers, the residual of the NEXT pass is
// requested before the stores of this one (vmcnt retires in order: a load queued behind stores waits for them), and the
// compiler's counted waits do the rest.
template <int NJ, int KT, int EPI>
__global__ void __launch_bounds__(512, 1) k_gemm_ws(GemmParams p, int np, int cpx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // ring depth: ten k-steps (3200 MFMA cycles of cover per wave); five where the residual double buffer needs the registers
    constexpr int BN = NJ * 16, K = KT * 32, EP_LD = BN + 4, RING = (EPI == 2 && NJ == 10) ? 5 : 10;
    constexpr int W_BYTES = KT * NJ * 1024;                    // [k-step][fragment][lane] x 16 B: a fragment read is lane-linear
    static_assert(KT % RING == 0, "the A ring is indexed by k-step mod RING");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, qx = blockIdx.x >> 3;
    if (qx >= cpx * np) return;
    const int panel = qx % np, chunk = xcd * cpx + qx / np;
    const int pbase = panel * BN;
    // Panel column behind row i of fragment j.  The direct epilogues pair fragments (2u, 2u+1): accumulator row 4g + r of the
    // pair is column 32u + 8g + 4 (j & 1) + r, so a lane (g = lane >> 4) holds 8 consecutive columns = one 16-byte store and the
    // four lane groups 64 contiguous bytes of a row (32-byte pieces, one fragment at a time, were store-issue bound: 5000-6000
    // cycles per tile by the stamps).  An unpaired last fragment (NJ odd) and the shared epilogue keep the natural order.
    auto pcol = [](int j, int i) { return (EPI && NJ != 10 && j < (NJ & ~1)) ? (j >> 1) * 32 + (i >> 2) * 8 + (j & 1) * 4 + (i & 3) : 16 * j + i; };
    // W panel -> LDS: slot (kk, j, g, n16) holds W[pbase + 16 j + n16][32 kk + 8 g .. +8]
    {
        constexpr int NSLOT = KT * NJ * 64, PER = (NSLOT + 511) / 512;      // 13 pieces per thread, all in flight before the first store
        bf16x8_t wv[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int sidx = min(tid + 512 * u, NSLOT - 1);
            const int l = sidx & 63, f = sidx >> 6, j = f % NJ, kk = f / NJ;
            wv[u] = *reinterpret_cast<const bf16x8_t*>(p.w + (long long)(pbase + pcol(j, l & 15)) * K + 32 * kk + 8 * (l >> 4));
        }
#pragma unroll
        for (int u = 0; u < PER; ++u)
            if (tid + 512 * u < NSLOT) *reinterpret_cast<bf16x8_t*>(smem + (tid + 512 * u) * 16) = wv[u];
    }
    float* sbias = reinterpret_cast<float*>(smem + W_BYTES) + 8 * (8 * EP_LD);      // the panel's bias (zeros without one), fast epilogues
    if (EPI && tid < BN) sbias[tid] = p.bias ? p.bias[pbase + tid] : 0.f;
    __syncthreads();
    float* stage = reinterpret_cast<float*>(smem + W_BYTES) + wave * (8 * EP_LD);
    const int ttot = (int)((p.M + 31) / 32), nchunk = 8 * cpx, tpc = (ttot + nchunk - 1) / nchunk;
    const int t_begin = chunk * tpc, t_end = min(t_begin + tpc, ttot);
    const int l15 = lane & 15, q = lane >> 4;
    const int mlast = (int)p.M - 1;
    const bf16_t* abase = p.x0 + 8 * q;
    bf16x8_t areg[RING][2];
    // With the straight-line epilogues the ring loads are issued and waited for by hand: the compiler answers loads carried
    // around the tile loop with vmcnt(0) at the top of every tile.  Loads retire in order, so "slot kk has landed" = at most
    // as many operations outstanding as LOADS were issued after it: the 18 other ring loads plus, when the slot was filled
    // before the previous tile's epilogue, that epilogue's EOPS residual loads.
    // The two straight-line epilogues.  STAGED (160-column panels): 8-row passes through the per-wave fp32 buffer, every lane
    // then writes 16-byte cells of whole rows (12 stores per tile, full lines).  Direct (80-column panels): fragment pairs as
    // 16-byte stores straight from the accumulators (pcol), 64 contiguous bytes per row and instruction.  Measured per shape:
    // 114688x320x320 39.7 / 38.5 us, + residual 47.4 / 56.8, x960 130 / 143; 28672x640x640 + residual 53.8 / 44.9.
    constexpr bool STAGED = NJ == 10;
    constexpr int EOPS = EPI != 2 ? 0 : (STAGED ? 4 * ((8 * (BN / 8) + 63) / 64) : 2 * ((NJ + 1) / 2));     // LOADS the epilogue issues (residual).  Its stores are NOT counted:
                                                                // loads retire in order among loads, but a store may retire ahead of an
                                                                // older load (seen: with stores in the count the ring was read too early
                                                                // once the epilogue got short), so the count must hold with every store
                                                                // already gone; while stores are pending it merely over-waits
    auto load_a = [&](int slot, int tile, int kk) {            // rows past M clamp to the last one (their results are never stored)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = min(tile * 32 + 16 * i + l15, mlast);
            const bf16_t* ap = abase + (long long)m * K + 32 * kk;
            if (EPI)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(areg[slot][i]) : "v"(ap) : "memory");
            else
                areg[slot][i] = *reinterpret_cast<const bf16x8_t*>(ap);
        }
    };
    // staged epilogue: this lane's cells of an 8-row pass (cell = lane + 64 c over 8 rows x BN / 8 columns-of-eight)
    constexpr int NC8 = BN / 8, NCELL = (8 * NC8 + 63) / 64;
    int crow[NCELL], ccol[NCELL];
    bool cval[NCELL];
#pragma unroll
    for (int c = 0; c < NCELL; ++c) {
        const int cell = lane + 64 * c;
        cval[c] = cell < 8 * NC8;
        crow[c] = cval[c] ? cell / NC8 : 0;
        ccol[c] = cval[c] ? (cell % NC8) * 8 : 0;
    }
    int t = t_begin + wave;
    if (t >= t_end) return;
#pragma unroll
    for (int s = 0; s < RING; ++s) load_a(s, t, s);
    // one tile; FIRST: no epilogue lies between the prologue's loads and this tile's k-steps (separate instantiations, not a
    // runtime flag: two wait statements tied to the same ring registers in two branches make the compiler copy those registers
    // ahead of the wait)
    auto tile_body = [&](auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const int tn = t + 8 < t_end ? t + 8 : t;              // the tile whose first k-steps refill the ring at the end of this one
        f32x4 acc[2][NJ];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
            const int slot = kk % RING;
            if (EPI) {
                if (kk < RING && !FIRST)
                    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(areg[slot][0]), "+v"(areg[slot][1]) : "n"(2 * (RING - 1) + EOPS) : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(areg[slot][0]), "+v"(areg[slot][1]) : "n"(2 * (RING - 1)) : "memory");
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const bf16x8_t wf = *reinterpret_cast<const bf16x8_t*>(smem + ((kk * NJ + j) * 64 + lane) * 16);
                acc[0][j] = mfma_16x16x32(wf, areg[slot][0], acc[0][j]);
                acc[1][j] = mfma_16x16x32(wf, areg[slot][1], acc[1][j]);
            }
            if (kk + RING < KT)
                load_a(slot, t, kk + RING);
            else
                load_a(slot, tn, kk + RING - KT);
            // W fragments: NJ / 2 - 1 reads ahead of the MFMAs that use them (left alone the scheduler keeps one pair in flight and
            // waits for an LDS round trip every four MFMAs)
            // (not in the residual variant: no difference there, 48-52 us either way)
            constexpr int AHEAD = NJ / 2 - 1;
            if (!(EPI == 2 && NJ == 10)) {
                __builtin_amdgcn_sched_group_barrier(0x100, AHEAD, 0);
#pragma unroll
                for (int g = 0; g < NJ - AHEAD; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * AHEAD, 0);
            }
            __builtin_amdgcn_sched_barrier(0);                 // keep each refill where its slot frees up (the scheduler sinks them all
        }                                                      // below the last MFMA otherwise)
        // epilogue: four passes of 8 rows; lanes whose row lies in the pass write their 16 columns-of-four, then every lane
        // takes 8-column cells of the staged rows (bias, residual, taps ... exactly as the tiled kernels)
        if (EPI && STAGED) {
            // residual cells: like the A ring, loaded and waited for by hand (a compiler-tracked load next to the stores makes
            // every wait a vmcnt(0): PMC showed these waves parked 57 % of their cycles).  The next pass's cells are requested
            // before this pass's stores; "pass p has landed" = at most the next pass's NCELL loads outstanding (loads retire in
            // order; pending stores only make the wait longer), the last pass has no younger load and drains.
            bf16x8_t rr[2][NCELL];
            auto load_res = [&](int pass) {
#pragma unroll
                for (int c = 0; c < NCELL; ++c) {
                    const bf16_t* rp = p.residual + (long long)min(t * 32 + 8 * pass + crow[c], mlast) * p.ldr + pbase + ccol[c];
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rr[pass & 1][c]) : "v"(rp) : "memory");
                }
            };
            if (EPI == 2) load_res(0);
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int i = pass >> 1, h = pass & 1;
                if ((l15 >> 3) == h) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) *reinterpret_cast<f32x4*>(stage + (l15 & 7) * EP_LD + 16 * j + 4 * q) = acc[i][j];
                }
                if (EPI == 2) {
                    if (pass < 3) {
                        load_res(pass + 1);
                        if (NCELL == 3)
                            asm volatile("s_waitcnt vmcnt(3)" : "+v"(rr[pass & 1][0]), "+v"(rr[pass & 1][1]), "+v"(rr[pass & 1][NCELL - 1])::"memory");
                        else
                            asm volatile("s_waitcnt vmcnt(2)" : "+v"(rr[pass & 1][0]), "+v"(rr[pass & 1][NCELL - 1])::"memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)" : "+v"(rr[pass & 1][0]), "+v"(rr[pass & 1][NCELL > 1 ? 1 : 0]), "+v"(rr[pass & 1][NCELL - 1])::"memory");
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int c = 0; c < NCELL; ++c) {
                    if (!cval[c]) continue;
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + crow[c] * EP_LD + ccol[c]);
                    const f32x4 hi4 = *reinterpret_cast<const f32x4*>(stage + crow[c] * EP_LD + ccol[c] + 4);
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(sbias + ccol[c]), b1 = *reinterpret_cast<const f32x4*>(sbias + ccol[c] + 4);
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = lo[e] + b0[e];
                        v[4 + e] = hi4[e] + b1[e];
                    }
                    if (EPI == 2) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += bf16_to_f32((bf16_t)rr[pass & 1][c][e]);
                    }
                    const u32x4 o = {pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7])};
                    const int m = t * 32 + 8 * pass + crow[c];
                    // compiler-visible stores (see the direct epilogue below for what an asm store did)
                    if (m <= mlast) *reinterpret_cast<u32x4*>(p.out + (long long)m * p.ldo + pbase + ccol[c]) = o;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
            return;
        }
        if (EPI) {
            // direct epilogue (no LDS round trip): fragment pairs as 16-byte stores, see pcol
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m = t * 32 + 16 * i + l15;
                const bf16_t* rrow = p.residual + (long long)min(m, mlast) * p.ldr + pbase;
                bf16_t* orow = p.out + (long long)m * p.ldo + pbase;
                constexpr int NP = NJ / 2;
                u32x4 rr[NP];
                u32x2 rl;
                if (EPI == 2) {
#pragma unroll
                    for (int u = 0; u < NP; ++u) rr[u] = *reinterpret_cast<const u32x4*>(rrow + 32 * u + 8 * q);
                    if (NJ & 1) rl = *reinterpret_cast<const u32x2*>(rrow + 16 * (NJ - 1) + 4 * q);
                }
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(sbias + 32 * u + 8 * q), b1 = *reinterpret_cast<const f32x4*>(sbias + 32 * u + 8 * q + 4);
                    f32x4 v0 = acc[i][2 * u] + b0, v1 = acc[i][2 * u + 1] + b1;
                    if (EPI == 2) {
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            v0[2 * e] += bf16_to_f32((bf16_t)(rr[u][e] & 0xffffu));
                            v0[2 * e + 1] += bf16_to_f32((bf16_t)(rr[u][e] >> 16));
                            v1[2 * e] += bf16_to_f32((bf16_t)(rr[u][2 + e] & 0xffffu));
                            v1[2 * e + 1] += bf16_to_f32((bf16_t)(rr[u][2 + e] >> 16));
                        }
                    }
                    const u32x4 o = {pack2_bf16(v0[0], v0[1]), pack2_bf16(v0[2], v0[3]), pack2_bf16(v1[0], v1[1]), pack2_bf16(v1[2], v1[3])};
                    // a compiler-visible store: issued from inline asm, the next LDS read reused its data registers and the rows
                    // 12-15 of every fragment went out with the first word overwritten (32 wait states after the store cured it;
                    // the compiler knows how long a store's data registers stay busy, an asm statement tells it nothing)
                    if (m <= mlast) *reinterpret_cast<u32x4*>(orow + 32 * u + 8 * q) = o;
                }
                if (NJ & 1) {
                    const f32x4 bj = *reinterpret_cast<const f32x4*>(sbias + 16 * (NJ - 1) + 4 * q);
                    f32x4 v = acc[i][NJ - 1] + bj;
                    if (EPI == 2) {
                        v[0] += bf16_to_f32((bf16_t)(rl[0] & 0xffffu));
                        v[1] += bf16_to_f32((bf16_t)(rl[0] >> 16));
                        v[2] += bf16_to_f32((bf16_t)(rl[1] & 0xffffu));
                        v[3] += bf16_to_f32((bf16_t)(rl[1] >> 16));
                    }
                    const u32x2 o = {pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3])};
                    if (m <= mlast) *reinterpret_cast<u32x2*>(orow + 16 * (NJ - 1) + 4 * q) = o;
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if ((l15 >> 3) == h) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) *reinterpret_cast<f32x4*>(stage + (l15 & 7) * EP_LD + 16 * j + 4 * q) = acc[i][j];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                epilogue_rows<NJ * 2, EP_LD, (8 * NJ * 2 + 63) / 64, true, false, false>(p, stage, t * 32 + 16 * i + 8 * h, 8, pbase, p.N, lane, 0, true, true);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
    };
    tile_body(std::true_type{});
    for (t += 8; t < t_end; t += 8) tile_body(std::false_type{});
}

extern "C" {

int vidseg_gemm_profiler_create(void** out) {
    VS_REQUIRE(out != nullptr, "gemm_profiler_create: null output");
    *out = new GemmProf();
    return VS_OK;
}

int vidseg_gemm_profiler_destroy(void* h) {
    GemmProf* gp = (GemmProf*)h;
    if (!gp) return VS_OK;
    // a recording profiler is referenced by its owner thread's t_prof: freeing it from another thread would leave that pointer dangling
    // (the owner's next launch_gemm reads g_prof.on through it).  The owner ends the recording first; on the owner itself destroy ends it.
    VS_REQUIRE(!gp->on || gp->owner == std::this_thread::get_id(),
               "gemm_profiler_destroy: the profiler is recording on another thread (call vidseg_gemm_profile_end there first)");
    gp->on = false;
    if (t_prof == gp) t_prof = &g_off;
    for (hipEvent_t e : gp->ev) (void)hipEventDestroy(e);
    delete gp;
    return VS_OK;
}

int vidseg_gemm_profile_begin(void* h) {
    VS_REQUIRE(h != nullptr, "gemm_profile_begin: null profiler");
    VS_REQUIRE(!((GemmProf*)h)->on || ((GemmProf*)h)->owner == std::this_thread::get_id(), "gemm_profile_begin: already recording on another thread");
    t_prof = (GemmProf*)h;
    g_prof.owner = std::this_thread::get_id();
    g_prof.on = true;
    g_prof.used = 0;
    g_prof.flops = 0.0;
    g_prof.launches = 0;
    g_prof.shapes.clear();
    return VS_OK;
}

// out[0] = total kernel milliseconds, out[1] = algorithmic FLOPs, out[2] = launches
int vidseg_gemm_profile_end(void* h, double* out) {
    VS_REQUIRE(h != nullptr, "gemm_profile_end: null profiler");
    GemmProf& gp = *(GemmProf*)h;
    VS_REQUIRE(!gp.on || gp.owner == std::this_thread::get_id(), "gemm_profile_end: call it on the thread that called vidseg_gemm_profile_begin");
    gp.on = false;
    if (t_prof == &gp) t_prof = &g_off;
    double ms = 0.0;
    for (int i = 0; i < 24; ++i) gp.kind_stats[i] = 0.0;
    for (int i = 0; i < 8; ++i) gp.kind_bytes[i] = 0.0;
    for (size_t i = 0; i + 1 < gp.used; i += 2) {
        float t = 0.f;
        hipError_t e = hipEventSynchronize(gp.ev[i + 1]);
        if (e == hipSuccess) e = hipEventElapsedTime(&t, gp.ev[i], gp.ev[i + 1]);
        if (e != hipSuccess) VS_FAIL(VS_ERR_HIP, "gemm_profile_end: %s", hipGetErrorString(e));
        ms += t;
        if (i / 2 < gp.shapes.size()) {
            const GemmProf::Shape& sh = gp.shapes[i / 2];
            const int kd = sh.kind >= 0 && sh.kind < 8 ? sh.kind : 0;
            gp.kind_stats[kd * 3] += t;
            gp.kind_stats[kd * 3 + 1] += 2.0 * (double)sh.M * (double)sh.N * (double)sh.K;
            gp.kind_stats[kd * 3 + 2] += 1.0;
            gp.kind_bytes[kd] += sh.alg_bytes;
        }
        if (knobs().shapes && i / 2 < gp.shapes.size()) {
            const GemmProf::Shape& sh = gp.shapes[i / 2];
            fprintf(stderr, "GEMMSHAPE M=%lld N=%d K=%d ks=%d up=%d st=%d act=%d split=%d us=%.1f kind=%d\n", sh.M, sh.N, sh.K, sh.ksize, sh.up,
                    sh.stride, sh.act, sh.ksplit, t * 1e3, sh.kind);
        }
    }
    out[0] = ms;
    out[1] = gp.flops;
    out[2] = (double)gp.launches;
    return VS_OK;
}

// Per-kernel split of the last profiled region: out[k*3 + {0,1,2}] = {milliseconds, algorithmic FLOPs, launches} for
// k = 0: k_gemm_dma (128x128), 1: k_gemm_ph big (256x320 / 256x256), 2: k_gemm_tile mid (128x320), 3: k_gemm_conv<256,64>,
// 4: k_gemm_p7 (224x320), 5: k_gemm_ws (weight-stationary streaming, short K), 6: k_gemm_p7x<5, false> (224x320 on split operands),
// 7: k_gemm_p7x<4, true> (224x256 on split operands, GEGLU epilogue).
int vidseg_gemm_profile_kinds(void* h, double* out) {
    VS_REQUIRE(h != nullptr, "gemm_profile_kinds: null profiler");
    for (int i = 0; i < 24; ++i) out[i] = ((GemmProf*)h)->kind_stats[i];
    return VS_OK;
}

// Algorithmic HBM bytes of the same region per kernel (same order): every operand once -- the activation tensor(s) the launch
// reads (the whole input image for a conv: the 9 taps re-read it through L1/L2, not through memory), the weight matrix, the
// residual, and every output it writes (16-bit result, fp32 result, fp16 taps); split-K partials are NOT algorithmic.
int vidseg_gemm_profile_bytes(void* h, double* out) {
    VS_REQUIRE(h != nullptr, "gemm_profile_bytes: null profiler");
    for (int i = 0; i < 8; ++i) out[i] = ((GemmProf*)h)->kind_bytes[i];
    return VS_OK;
}

// Split-K workspaces (caller-owned fp32 device memory), bound per (device, stream): two streams -- the two window lanes of
// pipeline.WindowPipeline, or the two devices of one process -- must not share partials, and nothing orders their launches.
// vidseg_bind_workspace registers the scratch the GEMMs launched on `stream` of the CURRENT device use (a null pointer unbinds); a
// launch on a stream without a binding runs without split-K.  The host keeps a scratch bound to a stream only while every launch that
// uses it is ordered on that stream (ops.workspace does: one Workspace per (device, torch stream)).  There is no process-wide default.
struct WsEntry {
    int dev;
    hipStream_t st;
    float* ws;
    long long floats;
};
static WsEntry g_ws_tab[64];
static int g_ws_n = 0;

int vidseg_bind_workspace(hipStream_t st, float* ws, long long floats) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    for (int i = 0; i < g_ws_n; ++i)
        if (g_ws_tab[i].dev == dev && g_ws_tab[i].st == st) {
            g_ws_tab[i].ws = ws;
            g_ws_tab[i].floats = ws ? floats : 0;
            return VS_OK;
        }
    VS_REQUIRE(g_ws_n < 64, "workspace table full (%d bindings)", g_ws_n);
    g_ws_tab[g_ws_n++] = WsEntry{dev, st, ws, ws ? floats : 0};
    return VS_OK;
}

static void ws_lookup(hipStream_t st, float*& ws, long long& floats) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    ws = nullptr;
    floats = 0;
    for (int i = 0; i < g_ws_n; ++i)
        if (g_ws_tab[i].dev == dev && g_ws_tab[i].st == st) {
            ws = g_ws_tab[i].ws;
            floats = g_ws_tab[i].floats;
            return;
        }
}

static int launch_gemm(const GemmParams& p_in, hipStream_t st) {
    GemmParams p = p_in;
    float* g_ws;
    long long g_ws_floats;
    ws_lookup(st, g_ws, g_ws_floats);
    VS_REQUIRE(p.K % BK == 0, "gemm: K=%d must be a multiple of %d", p.K, BK);
    VS_REQUIRE(p.C0 % BK == 0 && p.C1 % BK == 0, "gemm: source channels (%d,%d) must be multiples of %d", p.C0, p.C1, BK);
    VS_REQUIRE(p.M > 0 && p.N > 0, "gemm: empty problem");
    p.taps = p.ksize == 1 ? 1 : (p.tmode ? 3 : 9);
    p.kchunk = p.ksize == 1 ? p.C0 + p.C1 : 64;                          // conv weights are packed chunk-major (see GemmParams)
    p.a_fold = 0;
    if (p.split2) {                                                      // C0 = 3 Cin with Cin % 64 == 0 (C0 % 64 == 0 above and 3 is odd)
        VS_REQUIRE(!p.x1 && p.C1 == 0 && p.C0 % 3 == 0, "gemm: split operand images are single-source with 3 planes of channels (C0=%d)", p.C0);
        p.a_fold = 2 * (p.C0 / 3);
    }
    VS_REQUIRE(p.N % 8 == 0 && p.ldo % 8 == 0 && (!p.residual || p.ldr % 8 == 0) && (!p.tap || (p.tap_ld % 8 == 0 && p.tap_cols % 8 == 0)),
               "gemm: N=%d ldo=%d ldr=%d tap_ld=%d must be multiples of 8 (16-byte epilogue)", p.N, p.ldo, p.ldr, p.tap_ld);
    static VsOncePerDevice attr;
    if (attr.needs()) {
        (void)hipFuncSetAttribute((const void*)k_gemm_conv<128, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        (void)hipFuncSetAttribute((const void*)k_gemm_conv<256, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
        attr.mark();
    }
    const int force = knobs().tile;
    bool narrow = p.N <= 64 && p.act != 2 && p.M >= 256;
    if (force == 128) narrow = false;
    // Timing (bench.py roofline): by default the two events ride on the kernel's own dispatch packet (hipExtLaunchKernelGGL: start /
    // stop timestamps of the kernel, what rocprofv3 reports; with a split-K finish the stop event rides on the finish kernel).
    // VIDSEG_PROF_EXT=0: two hipEventRecord calls around the launch instead -- two extra barrier packets per launch, which cost the
    // window 1.7 ms and read ~9 us per launch more than the kernel trace.
    const int prof_ext = knobs().ext;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (g_prof.on) {
        ev0 = prof_event();
        ev1 = prof_event();
        if (!prof_ext) (void)hipEventRecord(ev0, st);
        g_prof.flops += 2.0 * (double)p.M * (double)p.N * (double)p.K;
        g_prof.launches++;
    }
    const bool ext = g_prof.on && prof_ext;
    auto launch = [&](auto kern, dim3 grid, unsigned block, size_t lds, auto... args) {
        if (ext)
            hipExtLaunchKernelGGL(kern, grid, dim3(block), (unsigned)lds, st, ev0, p.ksplit > 1 ? nullptr : ev1, 0, args...);
        else
            hipLaunchKernelGGL(kern, grid, dim3(block), (unsigned)lds, st, args...);
    };
    p.ksplit = 1;
    p.ws = nullptr;
    int kind = 0;                                              // 0: 128x128, 1: big, 2: mid, 3: narrow (profile log only)
    // weight-stationary streaming kernel: plain linears with K = 320 (160-column panels) or K = 640 (80-column panels) and enough
    // rows to give every wave of the 256 persistent blocks a few 32-row tiles; the 32 blocks of an XCD split into np panels x
    // cpx row chunks, and the launch is taken only when at most 2 of them stay idle.  VIDSEG_GEMM_WS=0 disables.
    const int ws_mode = knobs().ws;
    // k_gemm_ws is written for this chip: a fixed grid of 256 persistent blocks laid out over 8 XCDs and ~142 KB of dynamic LDS.  On
    // a device that does not offer that (fewer CUs, less LDS per block) or when the attribute call is refused, the launch goes to
    // the tiled kernels instead of failing (or idling half a larger chip).
    static VsPerDeviceFlag ws_flag;
    signed char& ws_fits = ws_flag.here();
    if (ws_fits < 0) {
        int dev = 0, cus = 0, lds = 0;
        ws_fits = hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
                  hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && cus == 256 &&
                  lds >= 10 * 10 * 1024 + 8 * 8 * 164 * 4 + 640;
        const size_t l320 = 10 * 10 * 1024 + 8 * 8 * 164 * 4 + 640, l640 = 5 * 20 * 1024 + 8 * 8 * 84 * 4 + 320;
        const void* fns[6] = {(const void*)k_gemm_ws<10, 10, 0>, (const void*)k_gemm_ws<10, 10, 1>, (const void*)k_gemm_ws<10, 10, 2>,
                              (const void*)k_gemm_ws<5, 20, 0>,  (const void*)k_gemm_ws<5, 20, 1>,  (const void*)k_gemm_ws<5, 20, 2>};
        for (int i = 0; i < 6 && ws_fits; ++i)
            if (hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)(i < 3 ? l320 : l640)) != hipSuccess) ws_fits = 0;
        (void)hipGetLastError();                                 // a refused attribute must not surface as this launch's error
    }
    if (p.plane_hi)
        VS_REQUIRE(p.ksize == 1 && p.plane_lo && p.act == 0 && p.plane_col0 % 8 == 0 && p.plane_ld % 8 == 0 && p.plane_col0 > 0 &&
                       p.plane_col0 < p.N && p.N - p.plane_col0 <= p.plane_ld && !p.residual,
                   "gemm: k | v planes need a plain linear, plane_col0 / plane_ld multiples of 8");
    if (p.blend) VS_REQUIRE(p.act != 2 && !p.plane_hi && (p.out_f32 || p.out_split3) && !p.out, "gemm: the blend epilogue belongs to fp32 / split-image results");
    if (p.out_split3 && p.act != 2)                            // a plain linear writing its consumer's operand image: any kernel below
        VS_REQUIRE(p.ksize == 1 && !p.out && !p.out_f32 && p.N % 8 == 0 && (p.ldo % 8) == 0, "gemm: split3 output needs a plain linear, N %% 8 == 0");
    if (p.out_split3 && p.act == 2) {                          // exact mode's GEGLU projection: always the 256 x 256 phased tile
        VS_REQUIRE(p.ksize == 1 && !p.out && !p.out_f32 && !p.residual && !p.tap && p.N % 64 == 0 && (p.ldo % 8) == 0,
                   "gemm: the GEGLU split3 output exists for the plain linear only");
        static VsOncePerDevice attr3;
        if (attr3.needs()) {
            (void)hipFuncSetAttribute((const void*)k_gemm_ph<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 + 256) * 128);
            attr3.mark();
        }
        p.gn = (p.N + 255) / 256 < 8 ? (p.N + 255) / 256 : 8;
        if (knobs().gg > 0) p.gn = knobs().gg < (p.N + 255) / 256 ? knobs().gg : (p.N + 255) / 256;   // A/B: panel width of the GEGLU tile order
        int kind3 = 1;
        if (p.geglu16) {                                       // weights interleaved in 16-row value | gate groups: the split tile (224 x 256)
            VS_REQUIRE(p.K % 192 == 0 && p.C0 == p.K && p.N % 256 == 0, "gemm: geglu16 needs K %% 192 == 0 and N %% 256 == 0 (K=%d N=%d)", p.K, p.N);
            static VsOncePerDevice attr4;
            if (attr4.needs()) {
                (void)hipFuncSetAttribute((const void*)k_gemm_p7x<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 + 256) * 128);
                attr4.mark();
            }
            const long long tiles4 = ((p.M + 223) / 224) * (p.N / 256);
            launch(k_gemm_p7x<4, true>, dim3((unsigned)tiles4), 512, 2 * (256 + 256) * 128, p);
            kind3 = 7;
        } else {
            const long long tiles3 = ((p.M + 255) / 256) * ((p.N + 255) / 256);
            launch(k_gemm_ph<4, true>, dim3((unsigned)tiles3), 512, 2 * (256 + 256) * 128, p);
        }
        if (g_prof.on) {
            if (!prof_ext) (void)hipEventRecord(ev1, st);
            g_prof.shapes.push_back({p.M, p.N, p.K, p.ksize, p.up, p.stride, p.act, 1, kind3, ((double)p.x0_bytes + 2.0 * p.N * p.K) * 2.0 / 3.0 + (VS_THIRD_PLANE(p.ldo) ? 6.0 : 4.0) * p.M * (p.N / 2)});
        }
        VS_CHECK_LAUNCH("gemm_geglu_split3");
        return VS_OK;
    }
    const int ws_bn = p.K == 320 ? 160 : 80;
    const int ws_np = p.N / ws_bn;
    const bool ws_ok = ws_mode && ws_fits && p.ksize == 1 && !p.x1 && p.C1 == 0 && p.C0 == p.K && (p.K == 320 || p.K == 640) && p.act != 2 &&
                       p.tmode == 0 && p.N % ws_bn == 0 && ws_np >= 1 && ws_np <= 32 && (32 % ws_np) <= 2 &&
                       !p.split2 && !p.res_f32 && !p.blend && !p.plane_hi && !p.out_split3 &&   // its epilogue has no exact-mode outputs
                      
                       (ws_mode == 2 || (p.M >= 16384 && p.K == 320));     // VIDSEG_GEMM_WS=2: whenever legal (tests); K = 640 is
                                                                            // slower than k_gemm_p7 so far (28672x640x640: 54 vs 44 us)
    if (ws_ok) {
        kind = 5;
        const int cpx = 32 / ws_np;
        const bool plain = p.out && !p.out_f32 && !p.rowvec && !p.rowadd && !p.tap && p.act == 0;
        const int epi = !plain ? 0 : (p.residual ? 2 : 1);
        const size_t l320 = 10 * 10 * 1024 + 8 * 8 * 164 * 4 + 640, l640 = 5 * 20 * 1024 + 8 * 8 * 84 * 4 + 320;
        if (p.K == 320) {
            if (epi == 0) launch(k_gemm_ws<10, 10, 0>, dim3(256), 512, l320, p, ws_np, cpx);
            else if (epi == 1) launch(k_gemm_ws<10, 10, 1>, dim3(256), 512, l320, p, ws_np, cpx);
            else launch(k_gemm_ws<10, 10, 2>, dim3(256), 512, l320, p, ws_np, cpx);
        } else {
            if (epi == 0) launch(k_gemm_ws<5, 20, 0>, dim3(256), 512, l640, p, ws_np, cpx);
            else if (epi == 1) launch(k_gemm_ws<5, 20, 1>, dim3(256), 512, l640, p, ws_np, cpx);
            else launch(k_gemm_ws<5, 20, 2>, dim3(256), 512, l640, p, ws_np, cpx);
        }
    } else if (narrow) {
        kind = 3;
        const long long tiles = ((p.M + 255) / 256) * ((p.N + 63) / 64);
        launch(k_gemm_conv<256, 64>, dim3((unsigned)tiles), 256, 2 * (256 + 64) * BK * 2, p);
    } else {
        const int nk = p.K / BK;
        const int nosplit = !knobs().split, use_dma = knobs().dma, big_mode = knobs().big, mid_mode = knobs().mid, ph_mode = knobs().ph,
                  p7_mode = knobs().p7, p7_phases = knobs().p7ph;
        constexpr int MID_LDS = 8 * 32 * 68 * 4;               // epilogue staging of 8 waves (69.6 KiB) > 2 x (128+320) x 64 B
        static VsOncePerDevice attr2;
        if (attr2.needs()) {
            attr2.mark();
            (void)hipFuncSetAttribute((const void*)k_gemm_dma<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 36864);
            (void)hipFuncSetAttribute((const void*)k_gemm_dma<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
            (void)hipFuncSetAttribute((const void*)k_gemm_tile<4, 4, 32, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, MID_LDS);
            (void)hipFuncSetAttribute((const void*)k_gemm_tile<5, 4, 32, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, MID_LDS);
            (void)hipFuncSetAttribute((const void*)k_gemm_tile<4, 4, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 + 256) * 128);
            (void)hipFuncSetAttribute((const void*)k_gemm_tile<5, 4, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 + 320) * 128);
            (void)hipFuncSetAttribute((const void*)k_gemm_ph<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 + 256) * 128);
            (void)hipFuncSetAttribute((const void*)k_gemm_ph<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 + 320) * 128);
            (void)hipFuncSetAttribute((const void*)k_gemm_p7<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 + 320) * 128);
            (void)hipFuncSetAttribute((const void*)k_gemm_p7<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 + 320) * 128);
            (void)hipFuncSetAttribute((const void*)k_gemm_p7x<5, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 + 320) * 128);
            (void)hipFuncSetAttribute((const void*)k_gemm_p7x<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 + 256) * 128);
        }
        // Blocks run in rounds over the resident slots, so the last round's fill decides the efficiency.  Pick the K
        // split that maximises fill / (1 + cost of writing+reading the fp32 partials); deterministic finish kernel.
        auto pick_split = [&](long long tiles, int slots, int min_nk = 40) {
            int bestS = 1;
            if (nosplit || nk < min_nk || p.act == 2 || !g_ws || tiles >= 4 * slots) return bestS;
            double best = 0.0;
            for (int S = 1; S <= 8; ++S) {
                if (S > 1 && (nk / S < 8 || (long long)S * p.M * p.N > g_ws_floats)) break;
                const long long items = tiles * S;
                const double fill = (double)items / (double)(((items + slots - 1) / slots) * slots);
                const double score = fill / (1.0 + (S > 1 ? 480.0 * S / (double)p.K : 0.0));
                if (score > best * 1.03) {
                    best = score;
                    bestS = S;
                }
            }
            return bestS;
        };
        const long long tiles = ((p.M + 127) / 128) * ((p.N + 127) / 128);
        // big tile: 256 x 320 when that divides N better (320/640/960/1280/1920 ...), else 256 x 256 (GEGLU needs pairs)
        const int NJ = (p.act != 2 && ((p.N + 319) / 320) * 320 <= ((p.N + 255) / 256) * 256) ? 5 : 4;
        const long long tiles_b = ((p.M + 255) / 256) * ((p.N + NJ * 64 - 1) / (NJ * 64));
        const long long tiles_mid = ((p.M + 127) / 128) * ((p.N + NJ * 64 - 1) / (NJ * 64));
        // measured (tools/shape_summary.py): the 128 x 320 8-wave tile only beats 128 x 128 on the 32x32-level projections
        // (28672 x 640 x 640: 61 -> 50 us); K = 320 layers and under-filled grids lose
        const bool mid_ok = tiles_mid >= 448 && p.K >= 640 && p.N <= 640 && p.act == 0;
        bool big = false, p7 = false, xsmall = false;
        int S = 1;
        if (big_mode && p.M >= 256) {
            S = pick_split(tiles_b, 256);
            long long items = tiles_b * S;
            double fill = (double)items / (double)(((items + 255) / 256) * 256);
            // the 224-row tile: useful work per occupied CU-round = grid fill x the rows of the last tile row that exist.  A window's
            // M = 28 * H * W makes it 1.0 where the 256-row tile gives 0.875 (see k_gemm_p7)
            if (p7_mode && ph_mode && NJ == 5 && p.act != 2) {
                const long long tm7 = (p.M + 223) / 224, tiles_7 = tm7 * ((p.N + 319) / 320);
                const bool x7ok = knobs().p7x && p.split2 && !p.x1 && p.C1 == 0 && p.K % 192 == 0 && p.C0 % 192 == 0;
                xsmall = x7ok && knobs().xsmall;
                const int S7 = pick_split(tiles_7, 256, xsmall ? 30 : 40);
                const long long items7 = tiles_7 * S7;
                const double fill7 = (double)items7 / (double)(((items7 + 255) / 256) * 256);
                const double eff7 = fill7 * (double)p.M / (double)(tm7 * 224);
                const double eff8 = fill * (double)p.M / (double)(((p.M + 255) / 256) * 256);
                // on split operands the 224-row tile has the native (hi, lo) staging (k_gemm_p7x: +13..17 % over the 3K walk, measured per
                // shape, profiles/r04_b_p7x_vs_p7.txt), the 256-row tile has not: the SVD window's M = 28 * 72 * 128 fills both heights
                // (with k_gemm_phx the 256-row tile has it too: the bonus applies only where phx is switched off, or with phx=2 for A/B runs)
                const double x7 = (x7ok && knobs().phx != 1) ? 1.15 : 1.0;
                if (eff7 * x7 > eff8 * 1.04 || p7_mode == 2) {
                    p7 = true;
                    S = S7;
                    items = items7;
                    fill = fill7;
                }
            }
            const long long tiles_sel = items / S;               // tile count of the chosen height
            // measured per shape (tools/shape_summary.py): the big tile wins once the K loop is long enough to amortise its
            // unoverlapped prologue/epilogue (one block per CU) and the grid fills the chip
            big = big_mode == 2 || (fill >= 0.70 && p.K >= 960 && (S == 1 || p.K / S >= 1440));
            // split operands on the 224-row tile (k_gemm_p7x): a K' / S of 960 is still five macro-tiles of three MFMA steps each, and
            // half a round of these tiles beats the 128 x 128 kernel's 3K walk -- the small-M launches of the pruned last step
            // (14 samples: 3584 / 896 rows at the 16^2 / 8^2 levels, 14336 x 640 at 32^2), measured per shape (tools/xsmall_bench.py)
            // 3584 x 1280 x 3840 83 -> 65 us, the 8^2-level convolutions of 14 samples 132 -> 118 / 253 -> 207 us; K' = 960 launches lose 3 us
            // and stay where they were (profiles/r04_i_xsmall_bench.txt; parity window 172.7 -> 171.3 ms in a same-box A/B)
            if (!big && p7 && xsmall && fill >= 0.5 && p.K >= 1920 && (S == 1 || p.K / S >= 960)) big = true;
            // with the rolled epilogue (10 us fixed cost per tile instead of 27) the big tile also takes the short-K layers whose
            // grid is at least a full round of 256 CUs: 114688x960x320 161 -> 151 us, 28672x1920x640 118 -> 109, 114688x320x640 96 -> 77;
            // GEGLU and small-M shapes still lose (measured per shape, tools/shape_summary.py)
            if (!big && big_mode == 1 && p.act != 2 && S == 1 && fill >= 0.85 && tiles_sel >= 200 && p.K >= 320 && (p.N >= 640 || p.K >= 640))
                big = true;
        }
        // panel width of the tile order: the `res` tiles resident on one XCD read (res/gn) A slabs and gn W slabs per pass
        const int panel_mode = knobs().panel;
        auto pick_gn = [&](int bm, int bn, int res, int split) {
            const int tn_all = (p.N + bn - 1) / bn;
            if (!panel_mode) return tn_all;
            if (knobs().gn > 0) return knobs().gn < tn_all ? knobs().gn : tn_all;                     // A/B: a fixed panel width
            const double a_slab = (double)bm * (p.K / p.taps) * (p.stride * p.stride) / (double)(p.up * p.up);   // input bytes/2 behind a tile row
            const double w_slab = (double)bn * p.K / split;
            int best = tn_all;
            double bc = 1e300;
            for (int g = 1; g <= tn_all; ++g) {
                const double c = (double)((res + g - 1) / g) * a_slab + (double)g * w_slab;
                if (c < bc * 0.999) {
                    bc = c;
                    best = g;
                }
            }
            return best;
        };
        if (big && p7) {
            p.ksplit = S;
            p.ws = S > 1 ? g_ws : nullptr;
            p.gn = pick_gn(224, 320, 32, S);
            const long long tiles_7 = ((p.M + 223) / 224) * ((p.N + 319) / 320);
            // split operand images (exact mode): every plane staged once per 64 original channels (k_gemm_p7x); p7x=0 keeps the plain
            // walk over the 3K axis
            if (knobs().p7x && p.split2 && !p.x1 && p.C1 == 0 && p.K % 192 == 0 && p.C0 % 192 == 0 && (p.K / 192) / S >= 1) {
                launch(k_gemm_p7x<5, false>, dim3((unsigned)(tiles_7 * S)), 512, 2 * (256 + 320) * 128, p);
                kind = 6;
            } else {
                if (p7_phases == 3)
                    launch(k_gemm_p7<3>, dim3((unsigned)(tiles_7 * S)), 512, 2 * (256 + 320) * 128, p);
                else
                    launch(k_gemm_p7<5>, dim3((unsigned)(tiles_7 * S)), 512, 2 * (256 + 320) * 128, p);
                kind = 4;
            }
        } else if (big) {
            p.ksplit = S;
            p.ws = S > 1 ? g_ws : nullptr;
            p.gn = pick_gn(256, NJ * 64, 32, S);
            if (ph_mode && NJ == 5 && knobs().phx != 0 && p.split2 && !p.x1 && p.C1 == 0 && p.K % 192 == 0 && p.C0 % 192 == 0 && (p.K / 192) / S >= 1 && p.act != 2) {
                static VsOncePerDevice attrx;
                if (attrx.needs()) {
                    (void)hipFuncSetAttribute((const void*)k_gemm_phx, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 + 320) * 128);
                    attrx.mark();
                }
                launch(k_gemm_phx, dim3((unsigned)(tiles_b * S)), 512, 2 * (256 + 320) * 128, p);   // split operands: each plane staged once
            } else if (ph_mode && NJ == 5)
                launch(k_gemm_ph<5>, dim3((unsigned)(tiles_b * S)), 512, 2 * (256 + 320) * 128, p);
            else if (ph_mode)
                launch(k_gemm_ph<4>, dim3((unsigned)(tiles_b * S)), 512, 2 * (256 + 256) * 128, p);
            else if (NJ == 5)
                launch(k_gemm_tile<5, 4, 64>, dim3((unsigned)(tiles_b * S)), 512, 2 * (256 + 320) * 128, p);
            else
                launch(k_gemm_tile<4, 4, 64>, dim3((unsigned)(tiles_b * S)), 512, 2 * (256 + 256) * 128, p);
            kind = 1;
        } else if (mid_mode && p.M >= 128 && (mid_mode == 2 || mid_ok)) {
            p.ksplit = 1;
            p.ws = nullptr;
            p.gn = pick_gn(128, NJ * 64, 64, 1);
            if (NJ == 5)
                launch(k_gemm_tile<5, 4, 32, 1>, dim3((unsigned)tiles_mid), 512, MID_LDS, p);
            else
                launch(k_gemm_tile<4, 4, 32, 1>, dim3((unsigned)tiles_mid), 512, MID_LDS, p);
            kind = 2;
        } else {
            S = pick_split(tiles, 512);
            p.ksplit = S;
            p.ws = S > 1 ? g_ws : nullptr;
            p.gn = pick_gn(128, 128, 128, S);
            // grids that do not fill the 4 x 256 block slots gain from two K-steps in flight per block (measured -10..-20 %);
            // full grids prefer the fourth resident block (VIDSEG_GEMM_DMA=3 forces the 3-stage variant everywhere)
            if (use_dma == 3 || (use_dma == 1 && tiles * S <= 512 && p.act != 2))
                launch(k_gemm_dma<3>, dim3((unsigned)(tiles * S)), 256, 49152, p);
            else if (use_dma)
                launch(k_gemm_dma<2>, dim3((unsigned)(tiles * S)), 256, 36864, p);
            else
                launch(k_gemm_conv<128, 128>, dim3((unsigned)(tiles * S)), 256, 2 * (128 + 128) * BK * 2, p);
        }
        if (p.ksplit > 1) {
            const long long n8 = p.M * (p.N / 8);
            if (ext)
                hipExtLaunchKernelGGL(k_splitk_finish, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, nullptr, ev1, 0, p);
            else
                k_splitk_finish<<<dim3((unsigned)((n8 + 255) / 256)), 256, 0, st>>>(p);
        }
    }
    if (g_prof.on) {
        if (!prof_ext) (void)hipEventRecord(ev1, st);
        const double n_out = p.act == 2 ? p.N / 2 : p.N;
        double ab = (double)p.x0_bytes + (double)p.x1_bytes + 2.0 * (double)p.N * (double)p.K;
        if (p.split2) ab *= 2.0 / 3.0;                           // two distinct planes per operand: (a_hi, a_lo) and (w_hi, w_lo)
        if (p.residual) ab += (p.res_f32 ? 4.0 : 2.0) * (double)p.M * n_out;
        if (p.blend && p.blend != reinterpret_cast<const float*>(p.residual)) ab += 4.0 * (double)p.M * n_out;
        if (p.out) ab += 2.0 * (double)p.M * n_out;
        if (p.out_f32) ab += 4.0 * (double)p.M * (p.plane_hi ? p.plane_col0 : n_out);
        if (p.plane_hi) ab += 4.0 * (double)p.M * (n_out - p.plane_col0);
        if (p.out_split3) ab += (VS_THIRD_PLANE(p.ldo) ? 6.0 : 4.0) * (double)p.M * n_out;
        if (p.tap) ab += 2.0 * (double)p.M * (double)p.tap_cols * (p.tap2 ? 2.0 : 1.0);
        g_prof.shapes.push_back({p.M, p.N, p.K, p.ksize, p.up, p.stride, p.act, p.ksplit, kind, ab});
    }
    VS_CHECK_LAUNCH("gemm_conv");
    return VS_OK;
}

// out[M][N] = A[M][K] @ W[N][K]^T (+bias)(+rowvec)(act)(+residual); A may be the channel concat of two [M][C] tensors.
int vidseg_linear_a16(const void* a0, const void* a1, int C0, int C1, long long M, const void* w, int N, const float* bias,
                       const float* rowvec, int rv_stride, int rows_per_sample, const void* residual, int ldr, void* out,
                       float* out_f32, int ldo, void* tap, void* tap2, int tap_cols, int tap_ld, const float* rowadd, int act,
                       hipStream_t st) {
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
    p.x0_bytes = M * p.C0 * 2;
    p.x1_bytes = M * p.C1 * 2;
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
    p.rowadd = rowadd;
    p.act = act;
    if (act == 2) VS_REQUIRE(N % 64 == 0 && out, "linear: GEGLU needs N %% 64 == 0 and a bf16 output");
    return launch_gemm(p, st);
}

// The exact mode's linear: vidseg_linear_a16 on a split operand image with an fp32 result and an fp32 residual added in the epilogue
// (t + to_out(attn(..)), x + proj_out(..): attention.py:636-757, 921-927 -- the residual stream stays fp32, no separate add pass).
int vidseg_linear_a16_rf32(const void* a, int K, long long M, const void* w, int N, const float* bias, const float* rowvec, int rv_stride,
                           int rows_per_sample, const float* residual_f32, int ldr, float* out_f32, int ldo, void* tap, void* tap2,
                           int tap_cols, int tap_ld, int act, const float* rowadd, hipStream_t st) {
    VS_REQUIRE(out_f32 != nullptr && act != 2 && (!residual_f32 || ldr % 8 == 0), "linear_rf32: needs an fp32 output, act != GEGLU, ldr %% 8 == 0");
    GemmParams p{};
    p.x0 = (const bf16_t*)a;
    p.C0 = K;
    p.ksize = 1;
    p.stride = 1;
    p.up = 1;
    p.Hin = p.Win = p.Hout = p.Wout = 1;
    p.w = (const bf16_t*)w;
    p.N = N;
    p.K = K;
    p.M = M;
    p.x0_bytes = M * K * 2;
    p.bias = bias;
    p.rowvec = rowvec;
    p.rv_stride = rv_stride;
    p.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1;
    p.residual = (const bf16_t*)residual_f32;
    p.res_f32 = 1;
    p.ldr = ldr;
    p.out_f32 = out_f32;
    p.ldo = ldo;
    p.tap = (f16*)tap;
    p.tap2 = (f16*)tap2;
    p.tap_cols = tap_cols;
    p.tap_ld = tap_ld;
    p.act = act;
    p.rowadd = rowadd;                                         // Step 4: lambda * mask on the rows of the modulated frames (ATT:646-663, VA:197-216)
    p.split2 = 1;                                              // the operands of this entry point ARE split images (header)
    return launch_gemm(p, st);
}

// vidseg_linear_a16_rf32 followed, inside the epilogue, by the VideoUNet's AlphaBlender (diffusionmodules/util.py:343-380 with
// image_only_indicator = 0): out = alpha * blend + (1 - alpha) * (a . w^T + bias + rowvec + residual).  The time stack's last linear
// (video_attention.py:281 -> :470-476) writes the mixed stream itself; the fp32 pass over three tensors that did it is gone.
int vidseg_linear_a16_rf32_blend(const void* a, int K, long long M, const void* w, int N, const float* bias, const float* residual_f32, int ldr,
                                 const float* blend_f32, float alpha, float* out_f32, hipStream_t st) {
    VS_REQUIRE(out_f32 != nullptr && blend_f32 != nullptr && (!residual_f32 || ldr % 8 == 0), "linear_rf32_blend: needs out, blend, ldr %% 8 == 0");
    GemmParams p{};
    p.x0 = (const bf16_t*)a;
    p.C0 = K;
    p.ksize = 1;
    p.stride = 1;
    p.up = 1;
    p.Hin = p.Win = p.Hout = p.Wout = 1;
    p.w = (const bf16_t*)w;
    p.N = N;
    p.K = K;
    p.M = M;
    p.x0_bytes = M * K * 2;
    p.bias = bias;
    p.rows_per_sample = 1;
    p.residual = (const bf16_t*)residual_f32;
    p.res_f32 = 1;
    p.ldr = ldr;
    p.out_f32 = out_f32;
    p.ldo = N;
    p.blend = blend_f32;
    p.blend_a = alpha;
    p.blend_b = 1.0f - alpha;
    p.split2 = 1;
    return launch_gemm(p, st);
}

// The fused q | k | v projection of a self-attention (attention.py:636-650; no bias there): columns [0, plane_col0) = q leave as fp32
// [M][ldo] (the attention kernel splits its own queries), columns [plane_col0, N) = k | v as the attention kernel's operand planes
// hi / lo fp16 [M][plane_ld] -- what vidseg_x_split_planes made of the fp32 k | v before, without their fp32 round trip.  Taps as usual.
int vidseg_linear_a16_qkv_planes(const void* a, int K, long long M, const void* w, int N, const float* bias, float* q_f32, int ldo,
                                 void* kv_hi, void* kv_lo, int plane_col0, int plane_ld, void* tap, void* tap2, int tap_cols, int tap_ld,
                                 hipStream_t st) {
    VS_REQUIRE(q_f32 != nullptr && kv_hi != nullptr && kv_lo != nullptr && ldo % 8 == 0 && ldo >= plane_col0,
               "linear_qkv_planes: needs q, both planes, ldo %% 8 == 0 and >= plane_col0");
    GemmParams p{};
    p.x0 = (const bf16_t*)a;
    p.C0 = K;
    p.ksize = 1;
    p.stride = 1;
    p.up = 1;
    p.Hin = p.Win = p.Hout = p.Wout = 1;
    p.w = (const bf16_t*)w;
    p.N = N;
    p.K = K;
    p.M = M;
    p.x0_bytes = M * K * 2;
    p.bias = bias;
    p.rows_per_sample = 1;
    p.out_f32 = q_f32;
    p.ldo = ldo;
    p.plane_hi = (f16*)kv_hi;
    p.plane_lo = (f16*)kv_lo;
    p.plane_col0 = plane_col0;
    p.plane_ld = plane_ld;
    p.tap = (f16*)tap;
    p.tap2 = (f16*)tap2;
    p.tap_cols = tap_cols;
    p.tap_ld = tap_ld;
    p.split2 = 1;
    return launch_gemm(p, st);
}

// The same linear writing [hi | lo | hi] of its fp32 result (fp16 [M][3 N]) instead of the fp32 tensor: for results whose only consumer
// is the next GEMM (the FF output projection feeding proj_out, attention.py:757 -> :921) -- saves the fp32 round trip and a split pass.
int vidseg_linear_a16_rf32_x3(const void* a, int K, long long M, const void* w, int N, const float* bias, const float* residual_f32, int ldr,
                              void* out_split3, hipStream_t st) {
    VS_REQUIRE(out_split3 != nullptr && N % 8 == 0 && (!residual_f32 || ldr % 8 == 0), "linear_rf32_x3: needs an output image, N %% 8 == 0, ldr %% 8 == 0");
    GemmParams p{};
    p.x0 = (const bf16_t*)a;
    p.C0 = K;
    p.ksize = 1;
    p.stride = 1;
    p.up = 1;
    p.Hin = p.Win = p.Hout = p.Wout = 1;
    p.w = (const bf16_t*)w;
    p.N = N;
    p.K = K;
    p.M = M;
    p.x0_bytes = M * K * 2;
    p.bias = bias;
    p.rows_per_sample = 1;
    p.residual = (const bf16_t*)residual_f32;
    p.res_f32 = 1;
    p.ldr = ldr;
    p.out_split3 = (f16*)out_split3;
    p.ldo = N;
    p.split2 = 1;
    return launch_gemm(p, st);
}

// The exact mode's GEGLU projection (attention.py:89-96) with the product formed in the epilogue and written as the FF output
// projection's split operand image: a split image [M][K] (K = 3 x the layer's width), w / bias GEGLU-interleaved like ops.pack_geglu,
// out_split3 fp16 [M][3 * (N / 2)] = [hi | lo | hi] of value * gelu_erf(gate) evaluated in fp32.
int vidseg_linear_a16_geglu_x3(const void* a, int K, long long M, const void* w, int N, const float* bias, void* out_split3, hipStream_t st) {
    VS_REQUIRE(out_split3 != nullptr && N % 64 == 0 && M >= 1, "linear_geglu_x3: N=%d M=%lld", N, M);
    GemmParams p{};
    p.x0 = (const bf16_t*)a;
    p.C0 = K;
    p.ksize = 1;
    p.stride = 1;
    p.up = 1;
    p.Hin = p.Win = p.Hout = p.Wout = 1;
    p.w = (const bf16_t*)w;
    p.N = N;
    p.K = K;
    p.M = M;
    p.x0_bytes = M * K * 2;
    p.bias = bias;
    p.rows_per_sample = 1;
    p.out_split3 = (f16*)out_split3;
    p.ldo = N / 2;
    p.act = 2;
    p.split2 = 1;                                              // a two-plane image: the 3 K walk of k_gemm_ph<4, true> folds (a_fold)
    return launch_gemm(p, st);
}

// The same projection on the split tile (k_gemm_p7x<4, true>: every plane of the operands staged once): w / bias interleaved in 16-ROW
// value | gate groups (exact.pack_geglu_x16) instead of 32-row ones; K %% 192 == 0 and N %% 256 == 0.
int vidseg_linear_a16_geglu_x3g16(const void* a, int K, long long M, const void* w, int N, const float* bias, void* out_split3, hipStream_t st) {
    VS_REQUIRE(out_split3 != nullptr && N % 256 == 0 && K % 192 == 0 && M >= 1, "linear_geglu_x3g16: N=%d K=%d M=%lld", N, K, M);
    GemmParams p{};
    p.x0 = (const bf16_t*)a;
    p.C0 = K;
    p.ksize = 1;
    p.stride = 1;
    p.up = 1;
    p.Hin = p.Win = p.Hout = p.Wout = 1;
    p.w = (const bf16_t*)w;
    p.N = N;
    p.K = K;
    p.M = M;
    p.x0_bytes = M * K * 2;
    p.bias = bias;
    p.rows_per_sample = 1;
    p.out_split3 = (f16*)out_split3;
    p.ldo = N / 2;
    p.act = 2;
    p.geglu16 = 1;
    p.split2 = 1;
    return launch_gemm(p, st);
}

// Same as vidseg_linear_a16 with the fp16 taps written in the temporal layout [(b s), t, c] (video_attention.py:152).
int vidseg_linear_a16_ttap(const void* a0, long long M, int C0, const void* w, int N, void* out, int ldo, void* tap, void* tap2,
                            int tap_cols, int tap_ld, int tap_T, int tap_S, hipStream_t st) {
    GemmParams p{};
    p.x0 = (const bf16_t*)a0;
    p.C0 = C0;
    p.ksize = 1;
    p.stride = 1;
    p.up = 1;
    p.Hin = p.Win = p.Hout = p.Wout = 1;
    p.w = (const bf16_t*)w;
    p.N = N;
    p.K = C0;
    p.M = M;
    p.x0_bytes = M * C0 * 2;
    p.rows_per_sample = 1;
    p.out = (bf16_t*)out;
    p.ldo = ldo;
    p.tap = (f16*)tap;
    p.tap2 = (f16*)tap2;
    p.tap_cols = tap_cols;
    p.tap_ld = tap_ld;
    p.tap_T = tap_T;
    p.tap_S = tap_S;
    return launch_gemm(p, st);
}

// Conv3d with kernel [3,1,1], padding [1,0,0] over frames (video_model.py:45-58): x NHWC bf16 [(b t)][HW][C],
// w packed [Cout][c/64][dt][c%64] (chunk-major K order, see GemmParams), + bias + per-sample emb vector + residual.
static int conv_temporal3_impl(const void* x, int C, int BT, int HW, int T, const void* w, int Cout, const float* bias,
                               const float* rowvec, int rv_stride, const void* residual, void* out, float* out_f32, hipStream_t st,
                               int split2 = 0, int res_f32 = 0, const float* blend = nullptr, float alpha = 0.f) {
    VS_REQUIRE(T >= 1 && BT % T == 0, "conv_temporal3: BT=%d T=%d", BT, T);
    GemmParams p{};
    p.x0 = (const bf16_t*)x;
    p.C0 = C;
    p.ksize = 3;
    p.tmode = 1;
    p.T = T;
    p.stride = 1;
    p.up = 1;
    p.Hin = p.Hout = HW;
    p.Win = p.Wout = 1;
    p.w = (const bf16_t*)w;
    p.N = Cout;
    p.K = 3 * C;
    p.M = (long long)BT * HW;
    p.x0_bytes = p.M * C * 2;
    p.bias = bias;
    p.rowvec = rowvec;
    p.rv_stride = rv_stride;
    p.rows_per_sample = HW;
    p.residual = (const bf16_t*)residual;
    p.ldr = Cout;
    p.out = (bf16_t*)out;
    p.out_f32 = out_f32;
    p.ldo = Cout;
    p.split2 = split2;
    p.res_f32 = res_f32;
    p.blend = blend;
    p.blend_a = alpha;
    p.blend_b = 1.0f - alpha;
    return launch_gemm(p, st);
}

int vidseg_conv_temporal3_a16(const void* x, int C, int BT, int HW, int T, const void* w, int Cout, const float* bias,
                               const float* rowvec, int rv_stride, const void* residual, void* out, hipStream_t st) {
    return conv_temporal3_impl(x, C, BT, HW, T, w, Cout, bias, rowvec, rv_stride, residual, out, nullptr, st);
}
// the same temporal convolution with the fp32 accumulators (+ bias + per-(b t) vector) stored as they are (exact mode, exact_ops.hip)
int vidseg_conv_temporal3_a16_f32(const void* x, int C, int BT, int HW, int T, const void* w, int Cout, const float* bias,
                                   const float* rowvec, int rv_stride, float* out_f32, hipStream_t st) {
    VS_REQUIRE(out_f32 != nullptr, "conv_temporal3_f32: output is null");
    return conv_temporal3_impl(x, C, BT, HW, T, w, Cout, bias, rowvec, rv_stride, nullptr, nullptr, out_f32, st, 1);   // split images (header)
}

// VideoResBlock's tail in one launch (video_model.py:66-89): the second [3,1,1] convolution of the time stack + its fp32 skip
// (`x + h`, openaimodel.py:369) + the AlphaBlender: out = alpha * blend + (1 - alpha) * (conv + bias + residual).
int vidseg_conv_temporal3_a16_f32_blend(const void* x, int C, int BT, int HW, int T, const void* w, int Cout, const float* bias,
                                         const float* residual_f32, const float* blend_f32, float alpha, float* out_f32, hipStream_t st) {
    VS_REQUIRE(out_f32 != nullptr && blend_f32 != nullptr, "conv_temporal3_f32_blend: needs out and blend");
    return conv_temporal3_impl(x, C, BT, HW, T, w, Cout, bias, nullptr, 0, residual_f32, nullptr, out_f32, st, 1, 1, blend_f32, alpha);
}

// 3x3 convolution, padding 1, NHWC bf16 activations, weight packed [Cout][c/64][kh*3+kw][c%64] (chunk-major K order).
static int conv3x3_impl(const void* x0, const void* x1, int C0, int C1, int B, int Hin, int Win, int stride, int up, const void* w,
                        int Cout, const float* bias, const float* rowvec, int rv_stride, const void* residual, void* out,
                        int pad, float* out_f32, void* tap, int tap_early, hipStream_t st, int res_f32 = 0, int split2 = 0) {
    VS_REQUIRE((stride == 1 || stride == 2) && (up == 1 || up == 2) && (pad == 0 || pad == 1), "conv3x3: stride=%d up=%d pad=%d", stride,
               up, pad);
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
    p.pad = pad;
    p.out_f32 = out_f32;
    p.Hout = (Hin * up + 2 - 3) / stride + 1;
    p.Wout = (Win * up + 2 - 3) / stride + 1;
    p.w = (const bf16_t*)w;
    p.N = Cout;
    p.K = 9 * (p.C0 + p.C1);
    p.M = (long long)B * p.Hout * p.Wout;
    p.x0_bytes = (long long)B * Hin * Win * p.C0 * 2;
    p.x1_bytes = (long long)B * Hin * Win * p.C1 * 2;
    p.bias = bias;
    p.rowvec = rowvec;
    p.rv_stride = rv_stride;
    p.rows_per_sample = p.Hout * p.Wout;
    p.residual = (const bf16_t*)residual;
    p.res_f32 = res_f32;
    p.split2 = split2;
    p.ldr = Cout;
    p.out = (bf16_t*)out;
    p.ldo = Cout;
    if (tap) {
        p.tap = (f16*)tap;
        p.tap_cols = Cout;
        p.tap_ld = Cout;
        p.tap_early = tap_early ? 1 : 0;
    }
    return launch_gemm(p, st);
}

int vidseg_conv3x3_a16(const void* x0, const void* x1, int C0, int C1, int B, int Hin, int Win, int stride, int up, const void* w,
                        int Cout, const float* bias, const float* rowvec, int rv_stride, const void* residual, void* out,
                        int pad, float* out_f32, hipStream_t st) {
    return conv3x3_impl(x0, x1, C0, C1, B, Hin, Win, stride, up, w, Cout, bias, rowvec, rv_stride, residual, out, pad, out_f32, nullptr, 0, st);
}

// The exact mode's 3x3 conv: split operand image in, fp32 out, fp32 residual (the ResBlock's `skip_connection(x) + h`, openaimodel.py:369)
// added in the epilogue.
int vidseg_conv3x3_a16_rf32(const void* x, int C, int B, int Hin, int Win, int stride, int up, const void* w, int Cout, const float* bias,
                            const float* rowvec, int rv_stride, const float* residual_f32, float* out_f32, hipStream_t st) {
    VS_REQUIRE(out_f32 != nullptr, "conv3x3_rf32: needs an fp32 output");
    return conv3x3_impl(x, nullptr, C, 0, B, Hin, Win, stride, up, w, Cout, bias, rowvec, rv_stride, residual_f32, nullptr, 1, out_f32, nullptr, 0,
                        st, 1, 1);
}

// The same convolution with an fp16 copy of the result taken inside the epilogue: tap_early = 1 after the bias and before the
// per-sample vector (ResBlock.in_layers_features, openaimodel.py:349-350), 0 after it and before the residual
// (ResBlock.out_layers_features, openaimodel.py:367-368).
int vidseg_conv3x3_a16_tap(const void* x0, const void* x1, int C0, int C1, int B, int Hin, int Win, int stride, int up, const void* w,
                            int Cout, const float* bias, const float* rowvec, int rv_stride, const void* residual, void* out,
                            int pad, void* tap_f16, int tap_early, hipStream_t st) {
    VS_REQUIRE(tap_f16 != nullptr, "conv3x3_tap: tap buffer is null");
    return conv3x3_impl(x0, x1, C0, C1, B, Hin, Win, stride, up, w, Cout, bias, rowvec, rv_stride, residual, out, pad, nullptr, tap_f16,
                        tap_early, st);
}

// Tiny-channel 3x3 convs.  conv_in: x NHWC fp32 [B][H][W][Cin], w fp32 [3][3][Cin][Cout] -> bf16 NHWC.
static int conv_in_impl(const float* x, const float* w, const float* bias, int B, int H, int W, int Cin, int Cout, void* out_bf16_nhwc,
                        float* out_f32_nhwc, hipStream_t st) {
    const long long npix = (long long)B * H * W;
    if (npix == 0 || Cout == 0) return VS_OK;
    VS_REQUIRE(Cout % 8 == 0 && Cout <= 2048 && 9 * Cin * Cout * 4 <= 160 * 1024 && W % 4 == 0, "conv_in: Cin=%d Cout=%d W=%d", Cin, Cout, W);
    const size_t lds = (size_t)9 * Cin * Cout * 4;
    static VsOncePerDevice attr;
    if (attr.needs()) {
        (void)hipFuncSetAttribute((const void*)k_conv_in, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr.mark();
    }
    const long long nquad = npix / 4;
    k_conv_in<<<dim3((unsigned)((nquad + CONV_IN_QUADS - 1) / CONV_IN_QUADS)), 256, lds, st>>>(x, w, bias, B, H, W, Cin, Cout,
                                                                                                (bf16_t*)out_bf16_nhwc, out_f32_nhwc);
    VS_CHECK_LAUNCH("conv_in");
    return VS_OK;
}

int vidseg_conv_in(const float* x, const float* w, const float* bias, int B, int H, int W, int Cin, int Cout, void* out_bf16_nhwc,
                   hipStream_t st) {
    return conv_in_impl(x, w, bias, B, H, W, Cin, Cout, out_bf16_nhwc, nullptr, st);
}
// the same convolution with the fp32 accumulators stored as they are (exact mode, exact_ops.hip)
int vidseg_conv_in_f32(const float* x, const float* w, const float* bias, int B, int H, int W, int Cin, int Cout, float* out_f32_nhwc,
                       hipStream_t st) {
    return conv_in_impl(x, w, bias, B, H, W, Cin, Cout, nullptr, out_f32_nhwc, st);
}

// conv_out: x NHWC bf16 [B][H][W][Cin], w bf16 [4][3][3][Cin] -> fp32 NCHW [B][4][H][W].
int vidseg_conv_out4(const void* x, const void* w, const float* bias, int B, int H, int W, int Cin, float* out_f32_nchw,
                     hipStream_t st) {
    VS_REQUIRE(Cin % 8 == 0, "conv_out4: Cin=%d must be a multiple of 8", Cin);
    const long long pix = (long long)B * H * W;
    if (pix == 0) return VS_OK;
    static int ws4 = -1;                                       // VIDSEG_CONV_OUT_WS=0: one wave per pixel, weights re-read per pixel
    if (ws4 < 0) ws4 = knobs().convout;
    const int npieces = 9 * (Cin / 8);
    const unsigned nblk = (unsigned)((pix + 3) / 4 < 2048 ? (pix + 3) / 4 : 2048);      // 8 blocks per CU walk the pixel list
    if (ws4 && pix >= 4096 && npieces <= 64 * 3)
        k_conv_out4_ws<3><<<dim3(nblk), 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)w, bias, B, H, W, Cin, out_f32_nchw);
    else if (ws4 && pix >= 4096 && npieces <= 64 * 6)
        k_conv_out4_ws<6><<<dim3(nblk), 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)w, bias, B, H, W, Cin, out_f32_nchw);
    else
        k_conv_out4<<<dim3((unsigned)((pix + 3) / 4)), 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)w, bias, B, H, W, Cin, out_f32_nchw);
    VS_CHECK_LAUNCH("conv_out4");
    return VS_OK;
}

}  // extern "C"
