// HIPCC_FLAGS: -ffp-contract=off
// (bit-faithful arithmetic: no implicit FMA contraction; every fused multiply-add below is an explicit fma())
// Post-UNet analysis kernels for gfx950: block aggregation, K-means (k-means++ / Lloyd, batched over
// the n_init restarts), 4-NN label propagation, dense cosine tracking and the trajectory vote.
//
// Arithmetic follows the reference bit for bit where it is defined (see include/vidseg_hip.h for the
// reference file:line each entry point replaces): fp16 roundings happen exactly where torch/numpy round,
// everything sklearn evaluates in float64 is evaluated in float64 here (v_fma_f64; inputs are fp16-valued
// so products are exact and sums are order-insensitive far below any decision threshold), reductions that
// feed later decisions use a fixed order (no floating-point atomics), ties follow the reference's rules.
#include "common.h"

// ---------------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ f16 f64_to_f16_rn(double d) {
    // correct single rounding double -> half: round-to-odd into f32, then RN to f16
    float r = __double2float_rz(d);
    if ((double)r != d) r = __uint_as_float(__float_as_uint(r) | 1u);
    return (f16)r;
}

// ---------------------------------------------------------------------------------------------
// a13 + a14 step (2): mean over decoder blocks (fp32 accumulate, one fp16 rounding) and per-token
// max-abs normalisation (fp16 / fp16 evaluated in fp32, one rounding).  One wave per token row.
// ---------------------------------------------------------------------------------------------
struct BlockPtrs { const f16* p[8]; };

__global__ void __launch_bounds__(256) k_mean_normalize(BlockPtrs blocks, int nblk, int64_t row0, int64_t rows, int C,
                                                        f16* __restrict__ out_mean, f16* __restrict__ out_norm) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t off = (row0 + row) * (int64_t)C;
    const float inv = (float)nblk;
    float vmax = 0.f;
    constexpr int MAXCH = 8;                       // chunks of 8 halves per lane: C <= 64*8*8 = 4096
    f16x8 m[MAXCH];
#pragma unroll
    for (int ch = 0; ch < MAXCH; ++ch) {
        const int c = lane * 8 + ch * 512;
        if (c < C) {
            float acc[8];
            f16x8 v = *reinterpret_cast<const f16x8*>(blocks.p[0] + off + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = (float)v[j];
            for (int b = 1; b < nblk; ++b) {
                v = *reinterpret_cast<const f16x8*>(blocks.p[b] + off + c);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = acc[j] + (float)v[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                f16 h = (f16)(acc[j] / inv);
                m[ch][j] = h;
                vmax = fmaxf(vmax, fabsf((float)h));
            }
        }
    }
    vmax = wave_max_f32(vmax);
#pragma unroll
    for (int ch = 0; ch < MAXCH; ++ch) {
        const int c = lane * 8 + ch * 512;
        if (c < C) {
            if (out_mean) *reinterpret_cast<f16x8*>(out_mean + row * (int64_t)C + c) = m[ch];
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (f16)((float)m[ch][j] / vmax);
            if (out_norm) *reinterpret_cast<f16x8*>(out_norm + row * (int64_t)C + c) = o;
        }
    }
}

// The same arithmetic for the shapes the windows have (nblk <= 4, C <= 1024), with every load of a row issued before the first
// value is used: with the block count and the chunk count known at compile time a lane has NB x NCH independent 16-byte loads in
// flight instead of walking them one round trip at a time (the generic kernel above: 56 us for the 73 MB of a configs[1] window =
// 1.3 TB/s, six dependent HBM latencies per wave).  Same order of fp32 operations, same roundings, bit-identical output.
template <int NB, int NCH, int R>
__global__ void __launch_bounds__(256) k_mean_normalize_fast(BlockPtrs blocks, int64_t row0, int64_t rows, int C, f16* __restrict__ out_mean,
                                                             f16* __restrict__ out_norm) {
    const int lane = threadIdx.x & 63;
    const int64_t rbase = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;      // R consecutive rows per wave: R x NB x NCH loads in flight
    if (rbase >= rows) return;
    const float inv = (float)NB;
    f16x8 v[R][NB][NCH];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t off = (row0 + rbase + r) * (int64_t)C;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int c = lane * 8 + ch * 512;
                if (c < C && rbase + r < rows) v[r][b][ch] = *reinterpret_cast<const f16x8*>(blocks.p[b] + off + c);
            }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = rbase + r;
        if (row >= rows) break;
        float vmax = 0.f;
        f16x8 m[NCH];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int c = lane * 8 + ch * 512;
            if (c < C) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float acc = (float)v[r][0][ch][j];
#pragma unroll
                    for (int b = 1; b < NB; ++b) acc = acc + (float)v[r][b][ch][j];
                    const f16 h = (f16)(acc / inv);
                    m[ch][j] = h;
                    vmax = fmaxf(vmax, fabsf((float)h));
                }
            }
        }
        vmax = wave_max_f32(vmax);
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int c = lane * 8 + ch * 512;
            if (c < C) {
                if (out_mean) *reinterpret_cast<f16x8*>(out_mean + row * (int64_t)C + c) = m[ch];
                f16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (f16)((float)m[ch][j] / vmax);
                if (out_norm) *reinterpret_cast<f16x8*>(out_norm + row * (int64_t)C + c) = o;
            }
        }
    }
}

// C % 128 == 0 (the 640-channel taps of decoder blocks 6-8): 16 lanes per row, four rows per wave, NQ = C / 128 chunks of 8 per lane.
// Every lane is busy on every load (the map above leaves 48 of 64 lanes idle on the second 512-channel chunk of a 640-channel row) and a
// wave has 4 x NB x NQ loads in flight.  The only cross-lane operation is the row's max |.|, which is exact in any order: same bits.
template <int NB, int NQ>
__global__ void __launch_bounds__(256) k_mean_normalize_rows16(BlockPtrs blocks, int64_t row0, int64_t rows, int C, f16* __restrict__ out_mean,
                                                               f16* __restrict__ out_norm) {
    const int lane = threadIdx.x & 63, sub = lane & 15;
    const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    const bool live = row < rows;
    const float inv = (float)NB;
    f16x8 v[NB][NQ];
    const int64_t off = (row0 + (live ? row : rows - 1)) * (int64_t)C;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int qd = 0; qd < NQ; ++qd) v[b][qd] = *reinterpret_cast<const f16x8*>(blocks.p[b] + off + (qd * 16 + sub) * 8);
    float vmax = 0.f;
    f16x8 m[NQ];
#pragma unroll
    for (int qd = 0; qd < NQ; ++qd) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float acc = (float)v[0][qd][j];
#pragma unroll
            for (int b = 1; b < NB; ++b) acc = acc + (float)v[b][qd][j];
            const f16 h = (f16)(acc / inv);
            m[qd][j] = h;
            vmax = fmaxf(vmax, fabsf((float)h));
        }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));     // within the row's 16 lanes
    if (!live) return;
#pragma unroll
    for (int qd = 0; qd < NQ; ++qd) {
        const int c = (qd * 16 + sub) * 8;
        if (out_mean) *reinterpret_cast<f16x8*>(out_mean + row * (int64_t)C + c) = m[qd];
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)((float)m[qd][j] / vmax);
        if (out_norm) *reinterpret_cast<f16x8*>(out_norm + row * (int64_t)C + c) = o;
    }
}

// ---------------------------------------------------------------------------------------------
// float64 64x64 tile product  D[s][j] = sum_c A[s][c] * B[j][c]  (256 threads, 4x4 per thread)
// ---------------------------------------------------------------------------------------------
#define TS 64
#define TJ 64
#define KC 32
#define LDP 65

struct RowsF16 {                 // rows of an fp16 matrix, optionally gathered and mean-centred
    const f16* base;
    const int32_t* gather;       // nullptr -> identity
    const double* mean;          // nullptr -> no centring
    int64_t nrows;               // logical rows (after gather)
    int C;
};
struct RowsF64 {
    const double* base;
    int64_t nrows;
    int C;
    const int32_t* gather = nullptr;   // optional row map (compacted restarts)
};

__device__ __forceinline__ void load_tile(double (*L)[LDP], const RowsF16& R, int64_t r0, int c0) {
    const int t = threadIdx.x;
    const int r = t >> 2, sub = (t & 3) * 8;
    const int64_t row = r0 + r;
    double v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.0;
    if (row < R.nrows && c0 + sub < R.C) {
        const int64_t src = R.gather ? (int64_t)R.gather[row] : row;
        f16x8 h = *reinterpret_cast<const f16x8*>(R.base + src * (int64_t)R.C + c0 + sub);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (double)h[j];
        if (R.mean) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] -= R.mean[c0 + sub + j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) L[sub + j][r] = v[j];
}
__device__ __forceinline__ void load_tile(double (*L)[LDP], const RowsF64& R, int64_t r0, int c0) {
    const int t = threadIdx.x;
    const int r = t >> 2, sub = (t & 3) * 8;
    const int64_t row = r0 + r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        double v = 0.0;
        if (row < R.nrows && c0 + sub + j < R.C) v = R.base[(R.gather ? (int64_t)R.gather[row] : row) * (int64_t)R.C + c0 + sub + j];
        L[sub + j][r] = v;
    }
}

// Tile product on the f64 matrix pipe: v_mfma_f64_16x16x4_f64, one 16-row slab per wave, four 16-column
// blocks.  acc[i][j] of lane l (wave w) is element  row = 16 w + (l >> 4) + 4 j,  col = 16 i + (l & 15).
typedef __attribute__((ext_vector_type(4))) double f64x4;
#define TROW(i, j) (((threadIdx.x >> 6) << 4) + ((threadIdx.x & 63) >> 4) + 4 * (j))
#define TCOL(i, j) (16 * (i) + (threadIdx.x & 15))

template <int NCB = 4>
__device__ __forceinline__ void tile_mma(f64x4 (&c)[4], double (*LA)[LDP], double (*LB)[LDP]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int k0 = 0; k0 < KC; k0 += 4) {
        const double a = LA[k0 + lk][w * 16 + lr];
#pragma unroll
        for (int i = 0; i < NCB; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, LB[k0 + lk][i * 16 + lr], c[i], 0, 0, 0);
    }
}

// Register-staged operand tiles: the global loads of K-chunk c+1 are issued before the MFMAs of chunk c (the row
// pointer / gather index is resolved once per tile), then converted and written to LDS after them.  Same values as
// load_tile, element for element.
struct TileRegsF16 {
    f16x8 h;
    double m[8];
    bool ok;
};
struct TileRegsF64 {
    double v[8];
    bool ok;
};
// Row pointer of this thread's tile row (thread t: row t>>2, 8 columns at (t&3)*8).  Out-of-range rows point at row 0
// so that every load below is unconditional (no divergent branches in the K loop); `ok` zeroes the values afterwards.
template <class T>
struct TileRow {
    const T* p;
    bool ok;
};
__device__ __forceinline__ TileRow<f16> tile_row_ptr(const RowsF16& R, int64_t r0) {
    const int64_t row = r0 + (threadIdx.x >> 2);
    const bool ok = row < R.nrows;
    const int64_t src = ok ? (R.gather ? (int64_t)R.gather[row] : row) : 0;
    return {R.base + src * (int64_t)R.C + (threadIdx.x & 3) * 8, ok};
}
__device__ __forceinline__ TileRow<double> tile_row_ptr(const RowsF64& R, int64_t r0) {
    const int64_t row = r0 + (threadIdx.x >> 2);
    const bool ok = row < R.nrows;
    const int64_t src = ok ? (R.gather ? (int64_t)R.gather[row] : row) : 0;
    return {R.base + src * (int64_t)R.C + (threadIdx.x & 3) * 8, ok};
}
// C is a multiple of 8 everywhere these tiles are used (16-byte fp16 rows); chunks past C are clamped to chunk 0.
__device__ __forceinline__ void tile_fetch(TileRegsF16& g, const RowsF16& R, const TileRow<f16>& rp, int c0) {
    const int sub = (threadIdx.x & 3) * 8;
    const bool inb = c0 + sub < R.C;
    const int cc = inb ? c0 : 0;
    g.ok = rp.ok && inb;
    g.h = *reinterpret_cast<const f16x8*>(rp.p + cc);
    if (R.mean) {
        const double* mp = R.mean + cc + sub;
#pragma unroll
        for (int j = 0; j < 8; ++j) g.m[j] = mp[j];
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) g.m[j] = 0.0;
    }
}
__device__ __forceinline__ void tile_fetch(TileRegsF64& g, const RowsF64& R, const TileRow<double>& rp, int c0) {
    const int sub = (threadIdx.x & 3) * 8;
    const bool inb = c0 + sub < R.C;
    const int cc = inb ? c0 : 0;
    g.ok = rp.ok && inb;
#pragma unroll
    for (int j = 0; j < 8; ++j) g.v[j] = rp.p[cc + j];
}
// Barrier that orders LDS traffic only: __syncthreads() carries a fence that makes hipcc drain the prefetched global
// loads (s_waitcnt vmcnt(0)) and with them the overlap this pipeline exists for.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void tile_commit(double (*L)[LDP], const TileRegsF16& g) {
    const int r = threadIdx.x >> 2, sub = (threadIdx.x & 3) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) L[sub + j][r] = g.ok ? (double)g.h[j] - g.m[j] : 0.0;
}
__device__ __forceinline__ void tile_commit(double (*L)[LDP], const TileRegsF64& g) {
    const int r = threadIdx.x >> 2, sub = (threadIdx.x & 3) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) L[sub + j][r] = g.ok ? g.v[j] : 0.0;
}
template <class R>
struct TileRegsOf;
template <>
struct TileRegsOf<RowsF16> {
    typedef TileRegsF16 type;
};
template <>
struct TileRegsOf<RowsF64> {
    typedef TileRegsF64 type;
};

template <int NCB = 4, class RA, class RB>
__device__ __forceinline__ void tile_gemm(double (&acc)[4][4], double (*LA)[LDP], double (*LB)[LDP], const RA& A, int64_t a0,
                                          const RB& B, int64_t b0, int C) {
    f64x4 c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = f64x4{0.0, 0.0, 0.0, 0.0};
    // two K-chunks in flight (named register sets, static indexing): chunk c+2 is requested while chunk c is multiplied
    typename TileRegsOf<RA>::type ga0, ga1;
    typename TileRegsOf<RB>::type gb0, gb1;
    const auto pa = tile_row_ptr(A, a0);
    const auto pb = tile_row_ptr(B, b0);
    tile_fetch(ga0, A, pa, 0);
    tile_fetch(gb0, B, pb, 0);
    if (KC < C) {
        tile_fetch(ga1, A, pa, KC);
        tile_fetch(gb1, B, pb, KC);
    }
    for (int c0 = 0; c0 < C; c0 += 2 * KC) {
        lds_barrier();                                   // the previous chunk's MFMAs are done with LDS
        tile_commit(LA, ga0);
        tile_commit(LB, gb0);
        if (c0 + 2 * KC < C) {
            tile_fetch(ga0, A, pa, c0 + 2 * KC);
            tile_fetch(gb0, B, pb, c0 + 2 * KC);
        }
        lds_barrier();
        tile_mma<NCB>(c, LA, LB);
        if (c0 + KC >= C) break;
        lds_barrier();
        tile_commit(LA, ga1);
        tile_commit(LB, gb1);
        if (c0 + 3 * KC < C) {
            tile_fetch(ga1, A, pa, c0 + 3 * KC);
            tile_fetch(gb1, B, pb, c0 + 3 * KC);
        }
        lds_barrier();
        tile_mma<NCB>(c, LA, LB);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = c[i][j];
}

// ---------------------------------------------------------------------------------------------
// K-means preparation: column mean (numpy axis-0 order: sequential over rows), squared row norms of
// the centred data, sum of per-feature variances (for sklearn's tol).
// ---------------------------------------------------------------------------------------------
// Column sums in two fixed-order levels (chunks of 256 rows summed in ascending row order, then the chunk
// partials in ascending chunk order): deterministic, differs from numpy's single sequential pass only in the
// last bits of a float64 sum of fp16 values, far below anything a decision depends on.
__global__ void __launch_bounds__(64) k_col_partial(const f16* __restrict__ x, const double* __restrict__ mean, int64_t n, int C,
                                                    int sq, double* __restrict__ part) {
    // grid (ceil(C/64), nchunk): part[chunk][c] = sum_i f(x[i][c]) over the chunk; f = identity or (x-mean)^2 / (x-mean)
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    const int64_t r0 = (int64_t)blockIdx.y * 256, r1 = min(n, r0 + 256);
    const double m = mean ? mean[c] : 0.0;
    double s = 0.0;
    for (int64_t i = r0; i < r1; ++i) {
        const double v = (double)x[i * C + c] - m;
        s += sq ? v * v : v;
    }
    part[(int64_t)blockIdx.y * C + c] = s;
}

__global__ void k_col_finish(const double* __restrict__ part, int nchunk, int C, double scale, const double* __restrict__ sub,
                             double sub_scale, double* __restrict__ out) {
    // out[c] = scale * sum_chunk part[chunk][c]  - sub_scale * sub[c]^2   (sub optional)
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int k = 0; k < nchunk; ++k) s += part[(int64_t)k * C + c];
    s *= scale;
    if (sub) s -= sub_scale * sub[c] * sub[c];
    out[c] = s;
}

__global__ void __launch_bounds__(256) k_row_sqnorm(const f16* __restrict__ x, const double* __restrict__ mean, int64_t n, int C,
                                                    double* __restrict__ xsq) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    double s = 0.0;
    for (int c = lane; c < C; c += 64) {
        double v = (double)x[row * C + c] - (mean ? mean[c] : 0.0);
        s = fma(v, v, s);
    }
    s = wave_sum_f64(s);
    if (lane == 0) xsq[row] = s;
}

// ---------------------------------------------------------------------------------------------
// k-means++ (sklearn cluster/_kmeans.py:174-274), all restarts in one launch.
//   kpp_dist : dcand[r][t][i] = min(closest[r][i], max(0, -2 x_cand.x_i + |x_cand|^2 + |x_i|^2)), per-tile pot partials
//   kpp_pick : finalise the previous step (first-min over trials), then scan closest[] and draw the next
//              candidates by searchsorted(cumsum(closest), u * pot)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_kpp_dist(RowsF16 X, const double* __restrict__ xsq, const int32_t* __restrict__ cand, int R,
                                                  int T, const double* __restrict__ closest, double* __restrict__ dcand,
                                                  double* __restrict__ part, int ntiles) {
    __shared__ double LA[KC][LDP];
    __shared__ double LB[KC][LDP];
    __shared__ double LD[TS][LDP];
    const int64_t n = X.nrows;
    const int J = R * T;
    const int64_t s0 = (int64_t)blockIdx.x * TS;
    const int j0 = blockIdx.y * TJ;
    RowsF16 B = X;
    B.gather = cand;
    B.nrows = J;
    double acc[4][4];
    tile_gemm(acc, LA, LB, X, s0, B, (int64_t)j0, X.C);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t s = s0 + TROW(i, j);
            const int col = j0 + TCOL(i, j);
            double v = 0.0;
            if (s < n && col < J) {
                const int r = col / T;
                double d = -2.0 * acc[i][j];
                d += xsq[cand[col]];
                d += xsq[s];
                d = fmax(d, 0.0);
                v = fmin(closest[(int64_t)r * n + s], d);
                dcand[(int64_t)col * n + s] = v;
            }
            LD[TROW(i, j)][TCOL(i, j)] = v;
        }
    __syncthreads();
    if (threadIdx.x < TJ && j0 + threadIdx.x < J) {
        double s = 0.0;
        for (int i = 0; i < TS; ++i) s += LD[i][threadIdx.x];
        part[(int64_t)(j0 + threadIdx.x) * ntiles + blockIdx.x] = s;
    }
}

__global__ void __launch_bounds__(1024) k_kpp_pick(int64_t n, int R, int K, int c, int Tprev, int Tnext, const double* __restrict__ u,
                                                   int ustride, double* __restrict__ closest, const double* __restrict__ dcand,
                                                   const double* __restrict__ part, int ntiles, double* __restrict__ pot,
                                                   const int32_t* __restrict__ cand_prev, int32_t* __restrict__ cand,
                                                   int32_t* __restrict__ center_ids, int Tmax, int staged) {
    // one block per restart r.  Finalises centre c-1 from the Tprev trial results, then (if c < K)
    // draws Tnext candidates for centre c.  The candidates of the previous round are READ from cand_prev [r * Tprev + k] and the new
    // ones WRITTEN to another buffer, cand [r * Tnext + k]: with one buffer, round 1 (Tprev = 1, Tnext = T) had block 0 write
    // cand[0 .. T) while blocks 1 .. T-1 had yet to read cand[1], cand[2], .. -- harmless while all R blocks start together on an idle
    // chip, a wrong first seed for a restart whose block starts late (another stream's kernels holding the CUs / the 112 KB of LDS):
    // found as run-to-run differences of whole restarts when the analysis overlaps the next window's UNet (tools/kmeans_race.py).  staged: the winning trial's n distances go through LDS (one coalesced read) and the
    // two per-thread segment walks read them there -- the same per-thread arithmetic as the direct walk (whose 112-byte-strided
    // global reads were the rest of this kernel's time).
    extern __shared__ double s_src[];
    const int r = blockIdx.x;
    __shared__ double s_pots[16];
    __shared__ int s_best;
    __shared__ double s_seg[1024];
    __shared__ int s_cnt[16];
    const int t = threadIdx.x;
    if ((t >> 6) < Tprev) {                           // wave k sums trial k's tile partials: lane-strided, then a butterfly
        const int k = t >> 6;                         // (four threads walking 224 partials one load at a time was a third of the kernel)
        double s = 0.0;
        const double* p = part + (int64_t)(r * Tprev + k) * ntiles;
        for (int i = t & 63; i < ntiles; i += 64) s += p[i];
        s = wave_sum_f64(s);
        if ((t & 63) == 0) s_pots[k] = s;
    }
    __syncthreads();
    if (t == 0) {
        int best = 0;
        for (int k = 1; k < Tprev; ++k)
            if (s_pots[k] < s_pots[best]) best = k;
        s_best = best;
        pot[r] = s_pots[best];
        center_ids[r * K + (c - 1)] = cand_prev[r * Tprev + best];
    }
    if (t < 16) s_cnt[t] = 0;
    __syncthreads();
    const int best = s_best;
    const double* src = dcand + (int64_t)(r * Tprev + best) * n;
    double* dst = closest + (int64_t)r * n;
    const int64_t seg = (n + 1023) / 1024;
    const int64_t i0 = (int64_t)t * seg, i1 = min(n, i0 + seg);
    double tot = 0.0;
    if (staged) {
        for (int64_t i = t; i < n; i += 1024) {
            const double v = src[i];
            s_src[i] = v;
            dst[i] = v;
        }
        __syncthreads();
        for (int64_t i = i0; i < i1; ++i) tot += s_src[i];
    } else {
        for (int64_t i = i0; i < i1; ++i) {
            double v = src[i];
            dst[i] = v;
            tot += v;
        }
    }
    if (c >= K) return;
    s_seg[t] = tot;
    __syncthreads();
    if (t < 64) {
        // exclusive scan of the 1024 segment totals by one wave: 16 consecutive totals per lane in order, the lane totals by a
        // log-step scan, the lane's offset added to its 16 prefixes.  (One thread walking all 1024 through LDS took most of this
        // kernel's 73 us.  Like the segment split itself this is another association of the same float64 sum than numpy's
        // sequential cumsum -- differences of one ulp of the running total, against thresholds u * pot drawn from a continuous
        // distribution; every reference fixture still picks the same candidates.)
        double loc[16];
        double run = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            loc[k] = run;
            run += s_seg[t * 16 + k];
        }
        double incl = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double up = __shfl_up(incl, o, 64);
            if (t >= o) incl += up;
        }
        const double base = incl - run;               // sum of the lanes before this one
#pragma unroll
        for (int k = 0; k < 16; ++k) s_seg[t * 16 + k] = base + loc[k];
    }
    __syncthreads();
    const double potr = s_pots[best];
    double rv[8];
    int cnt[8];
    for (int k = 0; k < Tnext; ++k) {
        rv[k] = u[(int64_t)r * ustride + k] * potr;
        cnt[k] = 0;
    }
    double run = s_seg[t];
    for (int64_t i = i0; i < i1; ++i) {
        run += staged ? s_src[i] : src[i];
        for (int k = 0; k < Tnext; ++k) cnt[k] += (run < rv[k]) ? 1 : 0;
    }
    for (int k = 0; k < Tnext; ++k)
        if (cnt[k]) atomicAdd(&s_cnt[k], cnt[k]);
    __syncthreads();
    if (t < Tnext) cand[r * Tnext + t] = (int32_t)min((int64_t)s_cnt[t], n - 1);
}

__global__ void k_gather_centers(RowsF16 X, const int32_t* __restrict__ ids, int J, double* __restrict__ centers) {
    const int j = blockIdx.x;
    if (j >= J) return;
    const int64_t src = ids[j];
    for (int c = threadIdx.x; c < X.C; c += blockDim.x)
        centers[(int64_t)j * X.C + c] = (double)X.base[src * X.C + c] - (X.mean ? X.mean[c] : 0.0);
}

// ---------------------------------------------------------------------------------------------
// Lloyd (sklearn cluster/_k_means_lloyd.pyx), restarts batched, `active` is a bit mask over restarts.
// ---------------------------------------------------------------------------------------------
__global__ void k_center_sqnorm(const double* __restrict__ centers, int J, int C, double* __restrict__ cn) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= J) return;
    double s = 0.0;
    for (int c = lane; c < C; c += 64) s = fma(centers[(int64_t)j * C + c], centers[(int64_t)j * C + c], s);
    s = wave_sum_f64(s);
    if (lane == 0) cn[j] = s;
}

__global__ void __launch_bounds__(256) k_lloyd_assign(RowsF16 X, const double* __restrict__ centers, const double* __restrict__ cn,
                                                      int K, int rpt, const unsigned* __restrict__ d_active,
                                                      const int32_t* __restrict__ slots, int nslots, const int32_t* __restrict__ colrow,
                                                      int32_t* __restrict__ labels, int32_t* __restrict__ changed) {
    // blockIdx.y selects `rpt` compacted restart slots (rpt*K <= 64 columns); slots[] lists the restarts that were still
    // running at the last host poll, d_active is the device truth.  argmin_j (|c_j|^2 - 2 x.c_j), first min wins.
    __shared__ double LA[KC][LDP];
    __shared__ double LB[KC][LDP];
    __shared__ double LD[TS][LDP];
    const unsigned active = *d_active;
    const int sbase = blockIdx.y * rpt;
    const int nloc = min(rpt, nslots - sbase);
    unsigned need = 0;
    for (int q = 0; q < nloc; ++q)
        if ((active >> slots[sbase + q]) & 1u) need = 1;
    if (!need) return;
    const int64_t n = X.nrows;
    const int64_t s0 = (int64_t)blockIdx.x * TS;
    const int ncol = nloc * K;
    RowsF64 B{centers, ncol, X.C, colrow + sbase * K};
    double acc[4][4];
    tile_gemm(acc, LA, LB, X, s0, B, 0, X.C);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = TCOL(i, j);
            double d = 0.0;
            if (col < ncol) d = cn[colrow[sbase * K + col]] + (-2.0 * acc[i][j]);
            LD[TROW(i, j)][col] = d;
        }
    __syncthreads();
    const int s = threadIdx.x & 63, q = threadIdx.x >> 6;      // 4 restarts handled per pass
    for (int qq = q; qq < nloc; qq += 4) {
        const int r = slots[sbase + qq];
        if (!((active >> r) & 1u) || s0 + s >= n) continue;
        const double* row = &LD[s][qq * K];
        double best = row[0];
        int lab = 0;
        for (int j = 1; j < K; ++j)
            if (row[j] < best) {
                best = row[j];
                lab = j;
            }
        const int64_t idx = (int64_t)r * n + s0 + s;
        if (labels[idx] != lab) atomicAdd(&changed[r], 1);
        labels[idx] = lab;
    }
}

__global__ void __launch_bounds__(256) k_lloyd_accum(RowsF16 X, const int32_t* __restrict__ labels, int R, int K,
                                                     const unsigned* __restrict__ d_active, const int32_t* __restrict__ slots, int S,
                                                     double* __restrict__ psum, int32_t* __restrict__ pcnt, int nblk) {
    // block (chunk, restart): fixed-order partial sums of the chunk's samples per cluster.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int r = slots[blockIdx.y];
    if (!((*d_active >> r) & 1u)) return;
    const int C = X.C;
    constexpr int CCH = 256;                                     // channels per pass, one per thread
    double* acc = reinterpret_cast<double*>(smem);               // [K][CCH]
    int* slab = reinterpret_cast<int*>(acc + (size_t)K * CCH);   // [S]
    int* scnt = slab + S;                                        // [K]
    const int64_t n = X.nrows;
    const int64_t s0 = (int64_t)blockIdx.x * S;
    const int ns = (int)min((int64_t)S, n - s0);
    for (int i = threadIdx.x; i < ns; i += 256) slab[i] = labels[(int64_t)r * n + s0 + i];
    for (int k = threadIdx.x; k < K; k += 256) scnt[k] = 0;
    __syncthreads();
    if (threadIdx.x == 0)
        for (int i = 0; i < ns; ++i) scnt[slab[i]]++;
    double* out = psum + ((int64_t)r * nblk + blockIdx.x) * K * C;
    for (int c0 = 0; c0 < C; c0 += CCH) {
        const int c = c0 + threadIdx.x;
        for (int k = 0; k < K; ++k) acc[k * CCH + threadIdx.x] = 0.0;
        if (c < C) {
            const double m = X.mean ? X.mean[c] : 0.0;
            for (int i = 0; i < ns; ++i) {
                const double v = (double)X.base[(s0 + i) * C + c] - m;
                acc[slab[i] * CCH + threadIdx.x] += v;
            }
            for (int k = 0; k < K; ++k) out[(int64_t)k * C + c] = acc[k * CCH + threadIdx.x];
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += 256) pcnt[((int64_t)r * nblk + blockIdx.x) * K + k] = scnt[k];
}

__global__ void __launch_bounds__(256) k_lloyd_update(const double* __restrict__ psum, const int32_t* __restrict__ pcnt, int nblk,
                                                      int R, int K, int C, const unsigned* __restrict__ d_active, const int32_t* __restrict__ slots,
                                                      double* __restrict__ centers,
                                                      double* __restrict__ shift2, int32_t* __restrict__ counts) {
    // block (k, r): new centre = (sum over chunks, ascending) * (1/count); shift2 = |new-old|^2
    const int k = blockIdx.x, r = slots[blockIdx.y];
    if (!((*d_active >> r) & 1u)) return;
    __shared__ double red[256];
    int cnt = 0;
    for (int b = 0; b < nblk; ++b) cnt += pcnt[((int64_t)r * nblk + b) * K + k];
    double sh = 0.0;
    if (cnt > 0) {
        const double alpha = 1.0 / (double)cnt;
        for (int c = threadIdx.x; c < C; c += 256) {
            double s = 0.0;
            for (int b = 0; b < nblk; ++b) s += psum[(((int64_t)r * nblk + b) * K + k) * C + c];
            s *= alpha;
            const int64_t ci = ((int64_t)r * K + k) * C + c;
            const double d = s - centers[ci];
            sh = fma(d, d, sh);
            centers[ci] = s;
        }
    }
    red[threadIdx.x] = sh;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        shift2[r * K + k] = red[0];
        counts[r * K + k] = cnt;
    }
}

// Convergence bookkeeping of _kmeans_single_lloyd (cluster/_kmeans.py:715-734) on the device, so that a restart is
// frozen at exactly the iteration sklearn would stop it without a host round trip per iteration:
//   labels unchanged -> strict convergence;  else sum_k (sqrt(shift2_k))^2 <= tol -> tolerance convergence.
// state[0] = active mask, state[1] = strict mask, state[2] = error flags (bit 0: empty cluster seen), state[3+r] = n_iter.
__global__ void k_lloyd_status(int R, int K, int it, double tol, int32_t* __restrict__ changed, const double* __restrict__ shift2,
                               const int32_t* __restrict__ counts, unsigned* __restrict__ state) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned active = state[0], strict = state[1], err = state[2];
    for (int r = 0; r < R; ++r) {
        if (!((active >> r) & 1u)) continue;
        state[3 + r] = (unsigned)(it + 1);
        for (int k = 0; k < K; ++k)
            if (counts[r * K + k] == 0) err |= 1u;
        if (changed[r] == 0) {
            strict |= 1u << r;
            active &= ~(1u << r);
        } else {
            double tot = 0.0;
            for (int k = 0; k < K; ++k) {
                const double sh = sqrt(shift2[r * K + k]);
                tot += sh * sh;
            }
            if (tot <= tol) active &= ~(1u << r);
        }
        changed[r] = 0;
    }
    state[0] = active;
    state[1] = strict;
    state[2] = err;
}

// ---------------------------------------------------------------------------------------------
// Accelerated Lloyd iteration -- same labels as the plain E-step by construction, far less work per iteration:
//  (1) Hamerly-style bound filter: per (restart, sample) an upper bound ub on the distance to its own centre and a lower
//      bound lb on the distance to every other centre.  After the centres moved by delta_k, ub += delta_a and
//      lb -= max_{k != a} delta_k are still bounds (triangle inequality); if ub < lb with a safety margin nine orders of
//      magnitude above float64 rounding the argmin cannot have changed and the sample is skipped.
//  (2) the E-step proper runs on the gathered list of unsettled samples only (same MFMA tile arithmetic as
//      k_lloyd_assign, so the scores are bit-identical to the full E-step's) and refreshes ub/lb/labels, recording every
//      label change (sample, old, new).
//  (3) M-step on EXACT sums: sums[r][k][c] = sum of the raw fp16 values of the members.  fp16 values are multiples of
//      2^-24 below 2^16, so float64 adds/subtracts of up to 2^13 of them are exact in any order (2^20 for data in
//      [-1, 1] such as the max-normalised tokens); the sums are therefore maintained incrementally from the change
//      list (+x into the new cluster, -x out of the old one) and centre = (sum - count*mean) * (1/count).
//      sklearn's own reduction order depends on its OpenMP thread count; this one has no order at all.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_lloyd_filter(int64_t n, int K, const unsigned* __restrict__ d_active, const int32_t* __restrict__ slots,
                                                      const int32_t* __restrict__ labels, double* __restrict__ ub, double* __restrict__ lb,
                                                      const double* __restrict__ delta, const double* __restrict__ dtop,
                                                      int32_t* __restrict__ list, int32_t* __restrict__ nlist) {
    const int r = slots[blockIdx.y];
    if (!((*d_active >> r) & 1u)) return;
    const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    double u = 0.0, l = 1.0;                               // out-of-range lanes: "settled"
    if (i < n) {
        const int64_t idx = (int64_t)r * n + i;
        const int a = labels[idx];
        const double d1 = dtop[r * 3], d2 = dtop[r * 3 + 1];
        const int arg1 = (int)dtop[r * 3 + 2];
        u = ub[idx] + delta[r * K + a];
        l = lb[idx] - (a == arg1 ? d2 : d1);
        ub[idx] = u;
        lb[idx] = l;
    }
    // block-aggregated append: one atomic per 1024 samples (same-address atomics serialise at the memory side)
    __shared__ int wcnt[16], wbase[16];
    const bool need = !(u + 1e-9 * (u + l) < l);
    const unsigned long long bal = __ballot(need);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wcnt[wave] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < 16; ++w) {
            wbase[w] = tot;
            tot += wcnt[w];
        }
        const int base = tot ? atomicAdd(&nlist[r], tot) : 0;
        for (int w = 0; w < 16; ++w) wbase[w] += base;
    }
    __syncthreads();
    if (need) list[(int64_t)r * n + wbase[wave] + __popcll(bal & ((1ull << lane) - 1ull))] = (int32_t)i;
}

template <int NCB>
__global__ void __launch_bounds__(256) k_lloyd_assign_list(RowsF16 X, const double* __restrict__ centers, const double* __restrict__ cn,
                                                           const double* __restrict__ xsq, int K, const unsigned* __restrict__ d_active,
                                                           const int32_t* __restrict__ slots, const int32_t* __restrict__ list,
                                                           const int32_t* __restrict__ nlist, int32_t* __restrict__ labels,
                                                           double* __restrict__ ub, double* __restrict__ lb, int32_t* __restrict__ chg,
                                                           int32_t* __restrict__ changed) {
    __shared__ double LA[KC][LDP];
    __shared__ double LB[KC][LDP];
    __shared__ double LD[TS][LDP];
    const int r = slots[blockIdx.y];
    if (!((*d_active >> r) & 1u)) return;
    const int64_t n = X.nrows;
    const int m = nlist[r];
    const int64_t s0 = (int64_t)blockIdx.x * TS;
    if (s0 >= m) return;
    RowsF16 Xr = X;
    Xr.gather = list + (int64_t)r * n;
    Xr.nrows = m;
    RowsF64 B{centers + (int64_t)r * K * X.C, K, X.C};
    double acc[4][4];
    tile_gemm<NCB>(acc, LA, LB, Xr, s0, B, 0, X.C);
#pragma unroll
    for (int i = 0; i < NCB; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = TCOL(i, j);
            double d = 0.0;
            if (col < K) d = cn[r * K + col] + (-2.0 * acc[i][j]);
            LD[TROW(i, j)][col] = d;
        }
    __syncthreads();
    const int s = threadIdx.x;
    if (s < TS && s0 + s < m) {
        const int i = Xr.gather[s0 + s];
        const double* row = &LD[s][0];
        double best = row[0], second = INFINITY;
        int lab = 0;
        for (int j = 1; j < K; ++j) {
            const double v = row[j];
            if (v < best) {
                second = best;
                best = v;
                lab = j;
            } else if (v < second) {
                second = v;
            }
        }
        // conservative bounds: |error| of a computed squared distance is ~1e-13 (|x|^2 + |c|^2); widen by 1e-10 of that
        const double xs = xsq[i];
        const double e = 1e-10 * (xs + cn[r * K + lab]) + 1e-300;
        const int64_t idx = (int64_t)r * n + i;
        ub[idx] = sqrt(fmax(xs + best, 0.0) + e);
        lb[idx] = sqrt(fmax(xs + second - e, 0.0));
        const int old = labels[idx];
        if (old != lab) {
            const int pos = atomicAdd(&changed[r], 1);
            chg[((int64_t)r * n + pos) * 2] = i;
            chg[((int64_t)r * n + pos) * 2 + 1] = (old & 0xffff) | (lab << 16);
            labels[idx] = lab;
        }
    }
}

// M-step part 1: apply the change list to the exact raw sums.  Block (channel tile of 64, restart, entry split): its 8
// waves stream their share of the entries (one channel per lane, 4 entries in flight), accumulate the +-x deltas in
// one LDS array with ds_add_f64 and add the non-zero ones to the sums with float64 atomics.  Every partial value is an
// exact multiple of 2^-24, so neither the LDS nor the global accumulation order can change a bit of the result.
#define ACC_SPLIT 8
__global__ void __launch_bounds__(512) k_lloyd_accum_list(const f16* __restrict__ x, int64_t n, int C, int K, const unsigned* __restrict__ d_active,
                                                          const int32_t* __restrict__ slots, const int32_t* __restrict__ chg,
                                                          const int32_t* __restrict__ changed, double* __restrict__ sums,
                                                          int32_t* __restrict__ counts) {
    __shared__ double dl[64][64];                        // [k][channel in tile]
    __shared__ int dc[64];
    const int r = slots[blockIdx.y];
    if (!((*d_active >> r) & 1u)) return;
    const int m = changed[r];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int w0 = blockIdx.z * 8;                       // first global wave index of this block
    if (w0 * 4 >= m) return;                             // fewer entries than 4 per preceding wave: nothing left for this split
    const int c = blockIdx.x * 64 + lane;
    for (int i = threadIdx.x; i < K * 64; i += 512) (&dl[0][0])[i] = 0.0;
    if (threadIdx.x < 64) dc[threadIdx.x] = 0;
    __syncthreads();
    const int32_t* e = chg + (int64_t)r * n * 2;
    const bool cok = c < C;
    const bool cnt_block = blockIdx.x == 0 && lane == 0;
    for (int q0 = (w0 + wave) * 4; q0 < m; q0 += ACC_SPLIT * 8 * 4) {
        int idx[4], on[4];
        double v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bool ok = q0 + t < m;
            idx[t] = ok ? e[(q0 + t) * 2] : -1;
            on[t] = ok ? e[(q0 + t) * 2 + 1] : 0;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = (idx[t] >= 0 && cok) ? (double)x[(int64_t)idx[t] * C + c] : 0.0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (idx[t] < 0) continue;
            const int old = on[t] & 0xffff, nw = on[t] >> 16;
            if (cok) {
                __hip_atomic_fetch_add(&dl[nw][lane], v[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (old != 0xffff) __hip_atomic_fetch_add(&dl[old][lane], -v[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (cnt_block) {
                atomicAdd(&dc[nw], 1);
                if (old != 0xffff) atomicSub(&dc[old], 1);
            }
        }
    }
    __syncthreads();
    if (cok)
        for (int k = wave; k < K; k += 8) {
            const double d = dl[k][lane];
            if (d != 0.0) __hip_atomic_fetch_add(&sums[((int64_t)r * K + k) * C + c], d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    if (blockIdx.x == 0 && threadIdx.x < K && dc[threadIdx.x] != 0) atomicAdd(&counts[r * K + threadIdx.x], dc[threadIdx.x]);
}

// M-step part 2, block (k, restart): centre = (sum - count*mean) * (1/count) (_average_centers), shift2 = |new - old|^2
// (_center_shift), delta = sqrt(shift2) for the bound filter, |centre|^2 for the next E-step.  Relocation entries
// {new, sample, old} move one sample's raw values between the sums for this iteration only.  A cluster that is still
// empty (relocation bailed out) copies the biggest cluster exactly like _average_centers' in-place loop does: the
// averaged centre if the biggest cluster has a smaller id, its raw centred sum otherwise (_k_means_common.pyx:286-295).
__global__ void __launch_bounds__(256) k_lloyd_update_sums(const f16* __restrict__ x, const double* __restrict__ sums,
                                                           const int32_t* __restrict__ counts, const double* __restrict__ mean, int K, int C,
                                                           const unsigned* __restrict__ d_active, const int32_t* __restrict__ slots,
                                                           const int32_t* __restrict__ reloc, const int32_t* __restrict__ nreloc,
                                                           double* __restrict__ centers, double* __restrict__ shift2,
                                                           double* __restrict__ delta, double* __restrict__ cn) {
    const int k = blockIdx.x, r = slots[blockIdx.y];
    if (!((*d_active >> r) & 1u)) return;
    __shared__ double red[2][256];
    const int nr = nreloc[r];
    const int32_t* rl = reloc + r * 64 * 3;
    auto eff_count = [&](int j) {
        int c = counts[r * K + j];
        for (int e = 0; e < nr; ++e) c += (rl[e * 3] == j) - (rl[e * 3 + 2] == j);
        return c;
    };
    auto eff_sum = [&](int j, int c) {
        double v = sums[((int64_t)r * K + j) * C + c];
        for (int e = 0; e < nr; ++e) {
            if (rl[e * 3] == j) v += (double)x[(int64_t)rl[e * 3 + 1] * C + c];
            if (rl[e * 3 + 2] == j) v -= (double)x[(int64_t)rl[e * 3 + 1] * C + c];
        }
        return v;
    };
    const int cnt = eff_count(k);
    int src = k, scnt = cnt;
    bool averaged = true;
    if (cnt == 0) {                                         // np.argmax(weight_in_clusters): first maximum
        src = 0;
        scnt = eff_count(0);
        for (int j = 1; j < K; ++j) {
            const int cj = eff_count(j);
            if (cj > scnt) {
                scnt = cj;
                src = j;
            }
        }
        averaged = src < k;
    }
    double sh = 0.0, sq = 0.0;
    const double alpha = (scnt > 0 && averaged) ? 1.0 / (double)scnt : 1.0;
    for (int c = threadIdx.x; c < C; c += 256) {
        const int64_t ci = ((int64_t)r * K + k) * C + c;
        double v = centers[ci];
        if (scnt > 0) {
            double t = eff_sum(src, c) - (double)scnt * mean[c];
            if (averaged) t *= alpha;
            const double dd = t - v;
            sh = fma(dd, dd, sh);
            centers[ci] = t;
            v = t;
        }
        sq = fma(v, v, sq);
    }
    red[0][threadIdx.x] = sh;
    red[1][threadIdx.x] = sq;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            red[0][threadIdx.x] += red[0][threadIdx.x + o];
            red[1][threadIdx.x] += red[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        shift2[r * K + k] = red[0][0];
        delta[r * K + k] = sqrt(red[0][0]);
        cn[r * K + k] = red[1][0];
    }
}

// k_lloyd_status plus the bookkeeping of the accelerated path: the two largest centre movements per restart for the
// filter, list/change counters reset for the next iteration.
__global__ void __launch_bounds__(64) k_lloyd_status_list(int R, int K, int it, double tol, int32_t* __restrict__ changed,
                                                          const double* __restrict__ shift2, const int32_t* __restrict__ counts,
                                                          unsigned* __restrict__ state, const double* __restrict__ delta,
                                                          double* __restrict__ dtop, int32_t* __restrict__ nlist) {
    // one wave, lane r = restart r (R <= 29)
    const int r = threadIdx.x;
    const unsigned active = state[0];
    const bool mine = r < R && ((active >> r) & 1u);
    bool is_strict = false, stop = false, empty = false;
    if (mine) {
        state[3 + r] = (unsigned)(it + 1);
        double d1 = -1.0, d2 = -1.0, tot = 0.0;
        int a1 = 0;
        for (int k = 0; k < K; ++k) {
            const double sh = sqrt(shift2[r * K + k]);
            tot += sh * sh;
            const double d = delta[r * K + k];
            if (d > d1) {
                d2 = d1;
                d1 = d;
                a1 = k;
            } else if (d > d2) {
                d2 = d;
            }
        }
        if (changed[r] == 0) {
            is_strict = true;
            stop = true;
        } else if (tot <= tol) {
            stop = true;
        }
        dtop[r * 3] = d1;
        dtop[r * 3 + 1] = d2 < 0.0 ? d1 : d2;
        dtop[r * 3 + 2] = (double)a1;
        changed[r] = 0;
        nlist[r] = 0;
    }
    const unsigned m_strict = (unsigned)__ballot(is_strict), m_stop = (unsigned)__ballot(stop), m_empty = (unsigned)__ballot(empty);
    if (threadIdx.x == 0) {
        state[0] = active & ~m_stop;
        state[1] |= m_strict;
        if (m_empty) state[2] |= 1u;
    }
}

__global__ void __launch_bounds__(256) k_inertia(RowsF16 X, const double* __restrict__ centers, const int32_t* __restrict__ labels,
                                                 int R, int K, double* __restrict__ part, int nblk) {
    // block (chunk of 256 samples, restart): wave per sample, fixed-order sum inside the block
    __shared__ double ws[4];
    const int r = blockIdx.y, C = X.C;
    const int64_t n = X.nrows;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double tot = 0.0;
    for (int i = w; i < 256; i += 4) {
        const int64_t s = (int64_t)blockIdx.x * 256 + i;
        if (s >= n) break;
        const double* cp = centers + ((int64_t)r * K + labels[(int64_t)r * n + s]) * C;
        double q = 0.0;
        for (int c = lane; c < C; c += 64) {
            const double d = ((double)X.base[s * C + c] - (X.mean ? X.mean[c] : 0.0)) - cp[c];
            q = fma(d, d, q);
        }
        tot += wave_sum_f64(q);
    }
    if (lane == 0) ws[w] = tot;
    __syncthreads();
    if (threadIdx.x == 0) part[(int64_t)r * nblk + blockIdx.x] = ((ws[0] + ws[1]) + ws[2]) + ws[3];
}

__global__ void k_sum_rows(const double* __restrict__ part, int nblk, double* __restrict__ out) {
    const int r = blockIdx.x;
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int b = 0; b < nblk; ++b) s += part[(int64_t)r * nblk + b];
        out[r] = s;
    }
}

__global__ void k_add_mean(double* __restrict__ centers, const double* __restrict__ mean, int K, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < K * C) centers[i] += mean[i % C];
}

// ---------------------------------------------------------------------------------------------
// 4-NN classifier (sklearn neighbors, brute force in float64): d = |q|^2 - 2 q.y + |y|^2 clipped at 0
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int vote4(const int* lab, int nn) {
    // mode of the neighbours' classes, smallest class among the most frequent (sklearn _classification.py predict)
    int bestc = 0, bestl = 0;
    for (int p = 0; p < nn; ++p) {
        int c = 0;
        for (int q = 0; q < nn; ++q) c += (lab[q] == lab[p]);
        if (c > bestc || (c == bestc && lab[p] < bestl)) {
            bestc = c;
            bestl = lab[p];
        }
    }
    return bestl;
}

__global__ void k_vote4(const int32_t* __restrict__ idx, const int32_t* __restrict__ ylab, int64_t nq, int32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    int lab[4], nn = 0;
    for (int p = 0; p < 4; ++p)
        if (idx[i * 4 + p] >= 0) lab[nn++] = ylab[idx[i * 4 + p]];
    out[i] = vote4(lab, nn);
}

__global__ void __launch_bounds__(256) k_knn(RowsF16 Q, RowsF16 Y, const double* __restrict__ qq, const double* __restrict__ yy,
                                             const int32_t* __restrict__ ylab, int32_t* __restrict__ out,
                                             int32_t* __restrict__ out_idx) {
    __shared__ double LA[KC][LDP];
    __shared__ double LB[KC][LDP];
    __shared__ double LD[TS][LDP];
    const int64_t q0 = (int64_t)blockIdx.x * TS;
    double bd[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    int bi[4] = {-1, -1, -1, -1};
    for (int64_t y0 = 0; y0 < Y.nrows; y0 += TJ) {
        double acc[4][4];
        tile_gemm(acc, LA, LB, Q, q0, Y, y0, Q.C);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t qi = q0 + TROW(i, j), yi = y0 + TCOL(i, j);
                double d = INFINITY;
                if (qi < Q.nrows && yi < Y.nrows) {
                    d = -2.0 * acc[i][j];
                    d += qq[qi];
                    d += yy[yi];
                    d = fmax(d, 0.0);
                }
                LD[TROW(i, j)][TCOL(i, j)] = d;
            }
        __syncthreads();
        if (threadIdx.x < TS) {
            const double* row = LD[threadIdx.x];
            for (int j = 0; j < TJ; ++j) {
                const double d = row[j];
                if (d < bd[3]) {
                    int p = 3;
                    while (p > 0 && d < bd[p - 1]) {
                        bd[p] = bd[p - 1];
                        bi[p] = bi[p - 1];
                        --p;
                    }
                    bd[p] = d;
                    bi[p] = (int)(y0 + j);
                }
            }
        }
    }
    if (threadIdx.x < TS && q0 + threadIdx.x < Q.nrows) {
        if (out_idx)
            for (int p = 0; p < 4; ++p) out_idx[(q0 + threadIdx.x) * 4 + p] = bi[p];
        if (ylab) {
            int lab[4], nn = 0;
            for (int p = 0; p < 4; ++p)
                if (bi[p] >= 0) lab[nn++] = ylab[bi[p]];
            out[q0 + threadIdx.x] = vote4(lab, nn);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Dense tracking (feature_extraction.py:176-323)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_track_normalize(const f16* __restrict__ x, int64_t rows, int C, int nb, f16* __restrict__ out) {
    // out[b][row][:] = row normalised b+1 times: n = f16(sqrt(sum x^2)), x = f16(f32(x)/f32(n))
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    constexpr int MAXE = 32;                                     // C <= 2048
    f16 v[MAXE];
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const int c = lane + 64 * e;
        v[e] = (c < C) ? x[row * C + c] : (f16)0.f;
    }
    for (int b = 0; b < nb; ++b) {
        double s = 0.0;
#pragma unroll
        for (int e = 0; e < MAXE; ++e) {
            const double d = (double)v[e];
            s = fma(d, d, s);
        }
        s = wave_sum_f64(s);
        const float nrm = (float)f64_to_f16_rn(sqrt(s));
#pragma unroll
        for (int e = 0; e < MAXE; ++e) {
            const int c = lane + 64 * e;
            if (c < C) {
                v[e] = (f16)((float)v[e] / nrm);
                out[((int64_t)b * rows + row) * C + c] = v[e];
            }
        }
    }
}

__global__ void __launch_bounds__(256) k_track_cos(const f16* __restrict__ normed, int64_t FN, int N, int C, int f, int batch,
                                                   int tiles_per_batch, const int32_t* __restrict__ cur, int use_aux, f16 w1, f16 w2,
                                                   f16* __restrict__ blend) {
    // normed: [nb][F*N][C].  Query tile = 64 queries inside one 500-query batch b; targets = frame f+1
    // normalised b+1 times; aux = frame 0 normalised b+1 times.  blend[q][col] fp16.
    __shared__ double LA[KC][LDP];
    __shared__ double LB[KC][LDP];
    __shared__ double LD[KC][LDP];
    const int b = blockIdx.x / tiles_per_batch;
    const int tq = blockIdx.x % tiles_per_batch;
    const int q0 = b * batch + tq * TS;
    const int qend = min(N, (b + 1) * batch);
    if (q0 >= qend) return;
    const int t0 = blockIdx.y * TJ;
    RowsF16 A{normed + (int64_t)f * N * C, cur, nullptr, (int64_t)qend, C};                      // version 0, gathered
    RowsF16 T{normed + ((int64_t)b * FN + (int64_t)(f + 1) * N) * C, nullptr, nullptr, (int64_t)N, C};
    RowsF16 X{normed + ((int64_t)b * FN) * C, nullptr, nullptr, (int64_t)N, C};
    f64x4 cc[4], cx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cc[i] = cx[i] = f64x4{0.0, 0.0, 0.0, 0.0};
    for (int c0 = 0; c0 < C; c0 += KC) {
        __syncthreads();
        load_tile(LA, A, (int64_t)q0, c0);
        load_tile(LB, T, (int64_t)t0, c0);
        if (use_aux) load_tile(LD, X, (int64_t)t0, c0);
        __syncthreads();
        tile_mma(cc, LA, LB);
        if (use_aux) tile_mma(cx, LA, LD);
    }
    double acc[4][4], acx[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[i][j] = cc[i][j];
            acx[i][j] = cx[i][j];
        }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = q0 + TROW(i, j), col = t0 + TCOL(i, j);
            if (q < qend && col < N) {
                f16 c1 = f64_to_f16_rn(acc[i][j]);
                if (use_aux) {
                    const f16 c2 = f64_to_f16_rn(acx[i][j]);
                    const f16 p1 = (f16)((float)w1 * (float)c1);
                    const f16 p2 = (f16)((float)w2 * (float)c2);
                    c1 = (f16)((float)p1 + (float)p2);
                }
                blend[(int64_t)q * N + col] = c1;
            }
        }
}

// ---- numpy arg-introselect replay (oracle/npselect.py documents the algorithm) -------------------
template <class V, class I>
struct SelCtxT {
    typedef V value_t;
    const V* v;
    I* t;
};
typedef SelCtxT<float, short> SelCtx;           // fp16 rows widened to float, 16-bit indices (tracking)
typedef SelCtxT<double, int32_t> SelCtxD;       // float64 keys, 32-bit indices (empty-cluster relocation)
template <class S>
__device__ __forceinline__ typename S::value_t sv(const S& s, int i) {
    return s.v[s.t[i]];
}
template <class S>
__device__ __forceinline__ void ssw(const S& s, int a, int b) {
    auto x = s.t[a];
    s.t[a] = s.t[b];
    s.t[b] = x;
}
template <class S>
__device__ int sel_median5(const S& s, int o) {
    if (sv(s, o + 1) < sv(s, o + 0)) ssw(s, o + 1, o + 0);
    if (sv(s, o + 4) < sv(s, o + 3)) ssw(s, o + 4, o + 3);
    if (sv(s, o + 3) < sv(s, o + 0)) ssw(s, o + 3, o + 0);
    if (sv(s, o + 4) < sv(s, o + 1)) ssw(s, o + 4, o + 1);
    if (sv(s, o + 2) < sv(s, o + 1)) ssw(s, o + 2, o + 1);
    if (sv(s, o + 3) < sv(s, o + 2)) return (sv(s, o + 3) < sv(s, o + 1)) ? 1 : 3;
    return 2;
}
template <int LVL, class S>
__device__ void sel_introselect(const S& s, int off, int num, int kth) {
    int low = 0, high = num - 1;
    if (kth - low < 3) {
        for (int i = 0; i <= kth; ++i) {
            int minidx = i;
            auto minval = sv(s, off + i);
            for (int k = i + 1; k < num; ++k)
                if (sv(s, off + k) < minval) {
                    minidx = k;
                    minval = sv(s, off + k);
                }
            ssw(s, off + i, off + minidx);
        }
        return;
    }
    int depth = 0;
    for (int m = num; m > 1; m >>= 1) ++depth;
    depth *= 2;
    while (low + 1 < high) {
        int ll = low + 1, hh = high;
        if (depth > 0 || hh - ll < 5 || LVL >= 3) {
            const int mid = low + (high - low) / 2;
            if (sv(s, off + high) < sv(s, off + mid)) ssw(s, off + high, off + mid);
            if (sv(s, off + high) < sv(s, off + low)) ssw(s, off + high, off + low);
            if (sv(s, off + low) < sv(s, off + mid)) ssw(s, off + low, off + mid);
            ssw(s, off + mid, off + low + 1);
        } else {
            const int n2 = hh - ll, nmed = n2 / 5;
            int sub = 0;
            for (int i = 0; i < nmed; ++i, sub += 5) {
                const int m = sel_median5(s, off + ll + sub);
                ssw(s, off + ll + sub + m, off + ll + i);
            }
            if constexpr (LVL < 3) {
                if (nmed > 2) sel_introselect<LVL + 1>(s, off + ll, nmed, nmed / 2);
            }
            const int mid = ll + nmed / 2;
            ssw(s, off + mid, off + low);
            ll--;
            hh++;
        }
        depth--;
        const auto pivot = sv(s, off + low);
        for (;;) {
            do { ll++; } while (sv(s, off + ll) < pivot);
            do { hh--; } while (pivot < sv(s, off + hh));
            if (hh < ll) break;
            ssw(s, off + ll, off + hh);
        }
        ssw(s, off + low, off + hh);
        if (hh >= kth) high = hh - 1;
        if (hh <= kth) low = ll;
    }
    if (high == low + 1)
        if (sv(s, off + high) < sv(s, off + low)) ssw(s, off + high, off + low);
}

// Empty-cluster relocation (sklearn/cluster/_k_means_common.pyx:167-211), one block per running restart, a no-op unless
// the M-step left a cluster without members: squared distance of every sample to its own (old) centre, numpy's
// argpartition(distances, -n_empty) replayed on the device, the n_empty farthest samples (taken from the END of the
// partitioned index array, reversed) become the centres of the empty clusters (ascending cluster id) and leave their old
// clusters.  Labels are NOT touched (exactly like sklearn: the next E-step re-labels), so the exact member sums stay
// consistent; the adjustment lives in reloc[r] = {new cluster, sample, old cluster} and is applied by k_lloyd_update_sums.
__global__ void __launch_bounds__(256) k_lloyd_relocate(const f16* __restrict__ x, const double* __restrict__ mean, int64_t n, int C, int K,
                                                        const unsigned* __restrict__ d_active, const int32_t* __restrict__ slots,
                                                        const int32_t* __restrict__ labels, const double* __restrict__ centers,
                                                        const int32_t* __restrict__ counts, double* __restrict__ rd,
                                                        int32_t* __restrict__ rt, int32_t* __restrict__ reloc,
                                                        int32_t* __restrict__ nreloc) {
    const int r = slots[blockIdx.x];
    if (!((*d_active >> r) & 1u)) return;
    __shared__ int empties[64];
    __shared__ int n_empty;
    __shared__ double wmax[4];
    if (threadIdx.x == 0) {
        int ne = 0;
        for (int k = 0; k < K; ++k)
            if (counts[r * K + k] == 0) empties[ne++] = k;
        n_empty = ne;
        nreloc[r] = 0;
    }
    __syncthreads();
    if (n_empty == 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* d = rd + (int64_t)r * n;
    int32_t* t = rt + (int64_t)r * n;
    double mx = 0.0;
    for (int64_t i = wave; i < n; i += 4) {
        const double* cen = centers + ((int64_t)r * K + labels[(int64_t)r * n + i]) * C;
        double acc = 0.0;
        for (int c = lane; c < C; c += 64) {
            const double v = ((double)x[i * C + c] - mean[c]) - cen[c];
            acc = fma(v, v, acc);
        }
        acc = wave_sum_f64(acc);
        if (lane == 0) {
            d[i] = acc;
            t[i] = (int32_t)i;
        }
        mx = fmax(mx, acc);
    }
    if (lane == 0) wmax[wave] = mx;
    __syncthreads();
    if (threadIdx.x != 0) return;
    __threadfence_block();
    if (fmax(fmax(wmax[0], wmax[1]), fmax(wmax[2], wmax[3])) == 0.0) return;       // more clusters than distinct samples (:186-189)
    SelCtxD sc{d, t};
    sel_introselect<0>(sc, 0, (int)n, (int)n - n_empty);
    for (int idx = 0; idx < n_empty; ++idx) {
        const int far = t[n - 1 - idx];
        reloc[(r * 64 + idx) * 3 + 0] = empties[idx];
        reloc[(r * 64 + idx) * 3 + 1] = far;
        reloc[(r * 64 + idx) * 3 + 2] = labels[(int64_t)r * n + far];
    }
    nreloc[r] = n_empty;
}

__global__ void __launch_bounds__(64) k_row_select(const f16* __restrict__ blend, int N, int w, int32_t* __restrict__ next,
                                                   int32_t* __restrict__ tie_rows) {
    // one wave per query row: unique maximum -> its index; ties -> replay numpy's fp16 arg-introselect.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* v = reinterpret_cast<float*>(smem);
    short* t = reinterpret_cast<short*>(v + N);
    const int q = blockIdx.x, lane = threadIdx.x;
    float mx = -INFINITY;
    for (int i = lane; i < N; i += 64) {
        const float x = (float)blend[(int64_t)q * N + i];
        v[i] = x;
        t[i] = (short)i;
        mx = fmaxf(mx, x);
    }
    mx = wave_max_f32(mx);
    int cnt = 0, first = N;
    for (int i = lane; i < N; i += 64)
        if (v[i] == mx) {
            cnt++;
            first = min(first, i);
        }
    for (int o = 32; o > 0; o >>= 1) {
        cnt += __shfl_xor(cnt, o, 64);
        first = min(first, __shfl_xor(first, o, 64));
    }
    __syncthreads();
    if (lane == 0) {
        int res = first;
        if (cnt > 1) {
            SelCtx s{v, t};
            sel_introselect<0>(s, 0, N, N - 1);
            res = t[N - 1];
            if (tie_rows) atomicAdd(tie_rows, 1);
        }
        next[q] = res;
    }
}

// ---------------------------------------------------------------------------------------------
// Trajectory filter + majority vote + write-back (feature_extraction.py:392-421)
// ---------------------------------------------------------------------------------------------
__global__ void k_traj_vote(const int32_t* __restrict__ idx, const int32_t* __restrict__ labels, int F, int N, int w,
                            int spatial_filter, int32_t* __restrict__ common, int32_t* __restrict__ winner) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    bool keep = true;
    if (spatial_filter) {
        int ph = idx[p] / w, pw = idx[p] % w;
        for (int t = 1; t < F; ++t) {
            const int i = idx[t * N + p];
            const int h = i / w, ww = i % w;
            if (h - ph > 1 || ww - pw > 1) {
                keep = false;
                break;
            }
            ph = h;
            pw = ww;
        }
    }
    if (!keep) {
        common[p] = -1;
        return;
    }
    // Counter.most_common(1): highest count, ties -> label met first along the trajectory
    int best = 0, bestc = 0;
    for (int t = 0; t < F; ++t) {
        const int lt = labels[t * N + idx[t * N + p]];
        bool seen = false;
        for (int s = 0; s < t; ++s)
            if (labels[s * N + idx[s * N + p]] == lt) {
                seen = true;
                break;
            }
        if (seen) continue;
        int c = 0;
        for (int s = t; s < F; ++s) c += (labels[s * N + idx[s * N + p]] == lt);
        if (c > bestc) {
            bestc = c;
            best = lt;
        }
    }
    common[p] = best;
    for (int t = 0; t < F; ++t) atomicMax(&winner[t * N + idx[t * N + p]], p);   // last writer (largest p) wins
}

__global__ void k_traj_write(const int32_t* __restrict__ labels, const int32_t* __restrict__ common, const int32_t* __restrict__ winner,
                             int64_t total, int32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int wp = winner[i];
    out[i] = wp >= 0 ? common[wp] : labels[i];
}

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int vidseg_mean_normalize_f16(const void* const* blocks, int nblk, int64_t row0, int64_t rows, int C, void* out_mean, void* out_norm,
                              hipStream_t st) {
    VS_REQUIRE(nblk >= 1 && nblk <= 8, "mean_normalize: nblk=%d out of [1,8]", nblk);
    VS_REQUIRE(C % 8 == 0 && C >= 8 && C <= 4096, "mean_normalize: C=%d must be a multiple of 8 in [8,4096]", C);
    if (rows == 0) return VS_OK;
    BlockPtrs bp;
    for (int i = 0; i < nblk; ++i) bp.p[i] = static_cast<const f16*>(blocks[i]);
    f16* om = (f16*)out_mean;
    f16* on = (f16*)out_norm;
    const int nch = C <= 512 ? 1 : (C <= 1024 ? 2 : 0);
    constexpr int MN_R = 1;                                            // rows per wave of the fast kernel (2 measured 21.2 against 20.1 us: no gain)
    const dim3 grid((unsigned)cdiv64(rows, 4)), gridf((unsigned)cdiv64(rows, 4 * MN_R));
#define VS_MN_FAST(NB, NCH) k_mean_normalize_fast<NB, NCH, MN_R><<<gridf, 256, 0, st>>>(bp, row0, rows, C, om, on)
    static const bool rows16 = [] {                                    // VIDSEG_MN_ROWS16=0: the one-row-per-wave map (A/B; same bits)
        const char* e = getenv("VIDSEG_MN_ROWS16");
        return !(e && e[0] == '0');
    }();
    if (rows16 && C == 640 && nblk == 3 && rows > 0)
        k_mean_normalize_rows16<3, 5><<<dim3((unsigned)cdiv64(rows, 16)), 256, 0, st>>>(bp, row0, rows, C, om, on);
    else if (nch == 1 && nblk == 1) VS_MN_FAST(1, 1);
    else if (nch == 1 && nblk == 2) VS_MN_FAST(2, 1);
    else if (nch == 1 && nblk == 3) VS_MN_FAST(3, 1);
    else if (nch == 1 && nblk == 4) VS_MN_FAST(4, 1);
    else if (nch == 2 && nblk == 1) VS_MN_FAST(1, 2);
    else if (nch == 2 && nblk == 2) VS_MN_FAST(2, 2);
    else if (nch == 2 && nblk == 3) VS_MN_FAST(3, 2);
    else if (nch == 2 && nblk == 4) VS_MN_FAST(4, 2);
    else k_mean_normalize<<<grid, 256, 0, st>>>(bp, nblk, row0, rows, C, om, on);
#undef VS_MN_FAST
    VS_CHECK_LAUNCH("mean_normalize");
    return VS_OK;
}

int vidseg_kmeans_prepare(const void* x16, int64_t n, int C, double* mean, double* xsq, double* colvar, double* scratch,
                          hipStream_t st) {
    // scratch: 2 * ceil(n/256) * C doubles
    VS_REQUIRE(n > 0 && C % 8 == 0, "kmeans_prepare: n=%lld C=%d", (long long)n, C);
    const int nchunk = (int)cdiv64(n, 256);
    const dim3 g((C + 63) / 64, nchunk);
    double* p1 = scratch;
    double* p2 = scratch + (int64_t)nchunk * C;
    k_col_partial<<<g, 64, 0, st>>>((const f16*)x16, nullptr, n, C, 0, p1);
    k_col_finish<<<dim3((C + 63) / 64), 64, 0, st>>>(p1, nchunk, C, 1.0 / (double)n, nullptr, 0.0, mean);
    k_row_sqnorm<<<dim3((unsigned)cdiv64(n, 4)), 256, 0, st>>>((const f16*)x16, mean, n, C, xsq);
    if (colvar) {
        // var_c = mean_i (xc_ic - m2_c)^2 with m2 = mean of the centred column = E[xc^2] - m2^2 evaluated from two sums
        k_col_partial<<<g, 64, 0, st>>>((const f16*)x16, mean, n, C, 0, p1);
        k_col_finish<<<dim3((C + 63) / 64), 64, 0, st>>>(p1, nchunk, C, 1.0 / (double)n, nullptr, 0.0, colvar);   // m2 (tmp in colvar)
        k_col_partial<<<g, 64, 0, st>>>((const f16*)x16, mean, n, C, 1, p2);
        k_col_finish<<<dim3((C + 63) / 64), 64, 0, st>>>(p2, nchunk, C, 1.0 / (double)n, colvar, 1.0, colvar);
    }
    VS_CHECK_LAUNCH("kmeans_prepare");
    return VS_OK;
}

int vidseg_row_sqnorm_f64(const void* x16, int64_t n, int C, double* xsq, hipStream_t st) {
    if (n == 0) return VS_OK;
    k_row_sqnorm<<<dim3((unsigned)cdiv64(n, 4)), 256, 0, st>>>((const f16*)x16, nullptr, n, C, xsq);
    VS_CHECK_LAUNCH("row_sqnorm");
    return VS_OK;
}

// One k-means++ round for all R restarts: finalise centre c-1 (Tprev trials), draw Tnext candidates for
// centre c (if c < K) and evaluate them.  Call with c = 0 after writing cand[r] = first centre ids
// and closest = +inf (Tprev = 0, Tnext = 1 evaluates the first centre), then c = 1..K.
int vidseg_kpp_round_v2(const void* x16, const double* mean, const double* xsq, int64_t n, int C, int R, int K, int c, int Tprev,
                        int Tnext, int Tmax, const double* u /*[R][ustride], this round's uniforms*/, int ustride, double* closest,
                        double* dcand /*[R*Tmax][n]*/, double* part /*[R*Tmax][ntiles]*/, double* pot, int32_t* cand, int64_t cand_len,
                        int32_t* center_ids, hipStream_t st) {
    VS_REQUIRE(R >= 1 && R <= 32 && Tmax <= 8 && K >= 1, "kpp_round: R=%d Tmax=%d K=%d", R, Tmax, K);
    VS_REQUIRE(cand_len >= 2ll * R * Tmax, "kpp_round: cand holds %lld int32, needs 2 * R * Tmax = %d (two halves)", (long long)cand_len, 2 * R * Tmax);
    const int ntiles = (int)cdiv64(n, TS);
    RowsF16 X{(const f16*)x16, nullptr, mean, n, C};
    if (c > 0) {
        static VsPerDeviceFlag lds_flag;                               // 0: the device refused 120 KB of dynamic LDS -> distances stay in global memory
        signed char& lds_ok = lds_flag.here();
        if (lds_ok < 0) {
            lds_ok = hipFuncSetAttribute((const void*)k_kpp_pick, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024) == hipSuccess;
            (void)hipGetLastError();
        }
        const int staged = lds_ok && n * 8 <= 120 * 1024;              // 14336 tokens: 112 KB
        k_kpp_pick<<<dim3(R), 1024, staged ? (size_t)n * 8 : 0, st>>>(n, R, K, c, Tprev, Tnext, u, ustride, closest, dcand, part, ntiles, pot,
                                                                   cand + (size_t)((c - 1) & 1) * R * Tmax, cand + (size_t)(c & 1) * R * Tmax,
                                                                   center_ids, Tmax, staged);
        VS_CHECK_LAUNCH("kpp_pick");
    }
    cand += (size_t)(c & 1) * R * Tmax;                                // this round's candidates (round 0: the host's first picks in half 0)
    if (c < K) {
        // candidates of restart r live compactly at cand[r*Tnext + t]
        VS_REQUIRE(Tnext >= 1 && Tnext <= Tmax, "kpp_round: Tnext=%d out of [1,%d]", Tnext, Tmax);
        k_kpp_dist<<<dim3(ntiles, (R * Tnext + TJ - 1) / TJ), 256, 0, st>>>(X, xsq, cand, R, Tnext, closest, dcand, part, ntiles);
        VS_CHECK_LAUNCH("kpp_dist");
    }
    return VS_OK;
}

int vidseg_gather_rows_f64(const void* x16, const double* mean, int C, const int32_t* ids, int J, double* out, hipStream_t st) {
    RowsF16 X{(const f16*)x16, nullptr, mean, 0, C};
    k_gather_centers<<<dim3(J), 256, 0, st>>>(X, ids, J, out);
    VS_CHECK_LAUNCH("gather_rows");
    return VS_OK;
}

// One Lloyd iteration for the restarts in `active`: E-step (labels updated in place, changed[r] += #changes),
// M-step (fixed-order partial sums), centres updated in place, shift2[r][k] and counts[r][k] written.
int vidseg_lloyd_iter(const void* x16, const double* mean, int64_t n, int C, int R, int K, const unsigned* d_active,
                      const int32_t* slots, int nslots, const int32_t* colrow, int update_centers, double* centers, double* cnorm,
                      int32_t* labels, int32_t* changed, double* psum, int32_t* pcnt, int chunk, double* shift2, int32_t* counts,
                      hipStream_t st) {
    // slots[nslots]: restart ids to process (compacted by the host at its last poll); colrow[nslots*K]: centre row
    // (r*K + k) of every compacted column.
    VS_REQUIRE(K >= 1 && K <= 64 && R >= 1 && R <= 29 && nslots >= 1 && nslots <= R, "lloyd_iter: K=%d R=%d nslots=%d", K, R, nslots);
    RowsF16 X{(const f16*)x16, nullptr, mean, n, C};
    const int rpt = 64 / K;
    k_center_sqnorm<<<dim3((R * K + 3) / 4), 256, 0, st>>>(centers, R * K, C, cnorm);
    k_lloyd_assign<<<dim3((unsigned)cdiv64(n, TS), (nslots + rpt - 1) / rpt), 256, 0, st>>>(X, centers, cnorm, K, rpt, d_active, slots,
                                                                                         nslots, colrow, labels, changed);
    VS_CHECK_LAUNCH("lloyd_assign");
    if (update_centers) {
        const int nblk = (int)cdiv64(n, chunk);
        const size_t lds = (size_t)K * 256 * 8 + (size_t)chunk * 4 + (size_t)K * 4;
        VS_REQUIRE(lds <= 160 * 1024, "lloyd_iter: LDS %zu too large", lds);
        static VsOncePerDevice attr_set;
        if (attr_set.needs()) {
            (void)hipFuncSetAttribute((const void*)k_lloyd_accum, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set.mark();
        }
        k_lloyd_accum<<<dim3(nblk, nslots), 256, lds, st>>>(X, labels, R, K, d_active, slots, chunk, psum, pcnt, nblk);
        VS_CHECK_LAUNCH("lloyd_accum");
        k_lloyd_update<<<dim3(K, nslots), 256, 0, st>>>(psum, pcnt, nblk, R, K, C, d_active, slots, centers, shift2, counts);
        VS_CHECK_LAUNCH("lloyd_update");
    }
    return VS_OK;
}

// One accelerated Lloyd iteration (see the kernels above) including the convergence bookkeeping of vidseg_lloyd_status.
// Host-initialised state before it = 0: labels = -1, sums = 0, counts = 0, list[r] = 0..n-1, nlist[r] = n, changed = 0.
// Relocation scratch: reloc_d [R][n] f64, reloc_t [R][n] i32, reloc [R][64][3] i32, nreloc [R] i32.
int vidseg_lloyd_step(const void* x16, const double* mean, const double* xsq, int64_t n, int C, int R, int K, int it, double tol,
                      unsigned* state, const int32_t* slots, int nslots, double* centers, double* cnorm, double* sums, int32_t* counts,
                      int32_t* labels, double* ub, double* lb, int32_t* list, int32_t* nlist, int32_t* chg, int32_t* changed,
                      double* shift2, double* delta, double* dtop, double* reloc_d, int32_t* reloc_t, int32_t* reloc, int32_t* nreloc,
                      hipStream_t st) {
    VS_REQUIRE(K >= 1 && K <= 64 && R >= 1 && R <= 29 && nslots >= 1 && nslots <= R, "lloyd_step: K=%d R=%d nslots=%d", K, R, nslots);
    VS_REQUIRE(n < (1LL << 31) / 2, "lloyd_step: n=%lld too large", (long long)n);
    RowsF16 X{(const f16*)x16, nullptr, mean, n, C};
    const unsigned* d_active = state;
    if (it == 0)
        k_center_sqnorm<<<dim3((R * K + 3) / 4), 256, 0, st>>>(centers, R * K, C, cnorm);
    else
        k_lloyd_filter<<<dim3((unsigned)cdiv64(n, 1024), nslots), 1024, 0, st>>>(n, K, d_active, slots, labels, ub, lb, delta, dtop, list, nlist);
    const dim3 ga((unsigned)cdiv64(n, TS), nslots);
#define VS_ASSIGN_LIST(NCB) \
    k_lloyd_assign_list<NCB><<<ga, 256, 0, st>>>(X, centers, cnorm, xsq, K, d_active, slots, list, nlist, labels, ub, lb, chg, changed)
    switch ((K + 15) / 16) {
        case 1: VS_ASSIGN_LIST(1); break;
        case 2: VS_ASSIGN_LIST(2); break;
        case 3: VS_ASSIGN_LIST(3); break;
        default: VS_ASSIGN_LIST(4); break;
    }
#undef VS_ASSIGN_LIST
    VS_CHECK_LAUNCH("lloyd_assign_list");
    k_lloyd_accum_list<<<dim3((C + 63) / 64, nslots, ACC_SPLIT), 512, 0, st>>>((const f16*)x16, n, C, K, d_active, slots, chg, changed, sums, counts);
    VS_CHECK_LAUNCH("lloyd_accum_list");
    k_lloyd_relocate<<<dim3(nslots), 256, 0, st>>>((const f16*)x16, mean, n, C, K, d_active, slots, labels, centers, counts, reloc_d, reloc_t,
                                                   reloc, nreloc);
    VS_CHECK_LAUNCH("lloyd_relocate");
    k_lloyd_update_sums<<<dim3(K, nslots), 256, 0, st>>>((const f16*)x16, sums, counts, mean, K, C, d_active, slots, reloc, nreloc, centers,
                                                         shift2, delta, cnorm);
    VS_CHECK_LAUNCH("lloyd_update_sums");
    k_lloyd_status_list<<<dim3(1), 64, 0, st>>>(R, K, it, tol, changed, shift2, counts, state, delta, dtop, nlist);
    VS_CHECK_LAUNCH("lloyd_status_list");
    return VS_OK;
}

// state: device uint32 [3 + R] = {active mask, strict mask, error flags, n_iter[R]}; see k_lloyd_status.
int vidseg_lloyd_status(int R, int K, int it, double tol, int32_t* changed, const double* shift2, const int32_t* counts,
                        unsigned* state, hipStream_t st) {
    k_lloyd_status<<<dim3(1), 64, 0, st>>>(R, K, it, tol, changed, shift2, counts, state);
    VS_CHECK_LAUNCH("lloyd_status");
    return VS_OK;
}

int vidseg_kmeans_inertia(const void* x16, const double* mean, int64_t n, int C, int R, int K, const double* centers,
                          const int32_t* labels, double* part, double* inertia, hipStream_t st) {
    RowsF16 X{(const f16*)x16, nullptr, mean, n, C};
    const int nblk = (int)cdiv64(n, 256);
    k_inertia<<<dim3(nblk, R), 256, 0, st>>>(X, centers, labels, R, K, part, nblk);
    k_sum_rows<<<dim3(R), 64, 0, st>>>(part, nblk, inertia);
    VS_CHECK_LAUNCH("kmeans_inertia");
    return VS_OK;
}

int vidseg_add_mean_f64(double* centers, const double* mean, int K, int C, hipStream_t st) {
    k_add_mean<<<dim3((K * C + 255) / 256), 256, 0, st>>>(centers, mean, K, C);
    VS_CHECK_LAUNCH("add_mean");
    return VS_OK;
}

int vidseg_knn_vote(const void* q16, int64_t nq, const void* ref16, int64_t nref, int C, const double* qq, const double* yy,
                    const int32_t* ref_labels, int32_t* out, hipStream_t st) {
    VS_REQUIRE(C % 8 == 0, "knn_vote: C=%d must be a multiple of 8", C);
    if (nq == 0) return VS_OK;
    VS_REQUIRE(nref >= 1, "knn_vote: empty reference set");
    RowsF16 Q{(const f16*)q16, nullptr, nullptr, nq, C};
    RowsF16 Y{(const f16*)ref16, nullptr, nullptr, nref, C};
    k_knn<<<dim3((unsigned)cdiv64(nq, TS)), 256, 0, st>>>(Q, Y, qq, yy, ref_labels, out, nullptr);
    VS_CHECK_LAUNCH("knn_vote");
    return VS_OK;
}

// The two halves of the same classifier, split so that the distance search (label-independent) can run before the
// labels exist: top-4 neighbour indices, then the vote.
int vidseg_knn_top4(const void* q16, int64_t nq, const void* ref16, int64_t nref, int C, const double* qq, const double* yy,
                    int32_t* out_idx, hipStream_t st) {
    VS_REQUIRE(C % 8 == 0 && nref >= 1, "knn_top4: C=%d nref=%lld", C, (long long)nref);
    if (nq == 0) return VS_OK;
    RowsF16 Q{(const f16*)q16, nullptr, nullptr, nq, C};
    RowsF16 Y{(const f16*)ref16, nullptr, nullptr, nref, C};
    k_knn<<<dim3((unsigned)cdiv64(nq, TS)), 256, 0, st>>>(Q, Y, qq, yy, nullptr, nullptr, out_idx);
    VS_CHECK_LAUNCH("knn_top4");
    return VS_OK;
}

int vidseg_vote4(const int32_t* idx, const int32_t* ref_labels, int64_t nq, int32_t* out, hipStream_t st) {
    if (nq == 0) return VS_OK;
    k_vote4<<<dim3((unsigned)cdiv64(nq, 256)), 256, 0, st>>>(idx, ref_labels, nq, out);
    VS_CHECK_LAUNCH("vote4");
    return VS_OK;
}

int vidseg_track_normalize(const void* x16, int64_t rows, int C, int nb, void* out, hipStream_t st) {
    VS_REQUIRE(C <= 2048 && nb >= 1, "track_normalize: C=%d nb=%d", C, nb);
    if (rows == 0) return VS_OK;
    k_track_normalize<<<dim3((unsigned)cdiv64(rows, 4)), 256, 0, st>>>((const f16*)x16, rows, C, nb, (f16*)out);
    VS_CHECK_LAUNCH("track_normalize");
    return VS_OK;
}

// One frame pair f -> f+1: blend map then per-row arg-max with numpy's tie rule.
int vidseg_track_step(const void* normed, int F, int N, int w, int C, int f, int batch, const int32_t* cur, int use_aux, void* blend,
                      int32_t* next, int32_t* tie_rows, hipStream_t st) {
    VS_REQUIRE(C % 8 == 0 && N <= 32767 && f >= 0 && f + 1 < F, "track_step: C=%d N=%d f=%d F=%d", C, N, f, F);
    const int nb = N / batch + 1;
    const int tpb = (batch + TS - 1) / TS;
    const f16 w1 = (f16)((double)f / (double)(f + 1)), w2 = (f16)(1.0 / (double)(f + 1));
    k_track_cos<<<dim3(nb * tpb, (N + TJ - 1) / TJ), 256, 0, st>>>((const f16*)normed, (int64_t)F * N, N, C, f, batch, tpb, cur, use_aux,
                                                                   w1, w2, (f16*)blend);
    VS_CHECK_LAUNCH("track_cos");
    k_row_select<<<dim3(N), 64, (size_t)N * 6, st>>>((const f16*)blend, N, w, next, tie_rows);
    VS_CHECK_LAUNCH("row_select");
    return VS_OK;
}

int vidseg_trajectory_vote(const int32_t* idx, const int32_t* labels, int F, int N, int w, int spatial_filter, int32_t* common,
                           int32_t* winner, int32_t* out, hipStream_t st) {
    if (N == 0 || F == 0) return VS_OK;
    hipError_t e = hipMemsetAsync(winner, 0xFF, sizeof(int32_t) * (size_t)F * N, st);
    if (e != hipSuccess) VS_FAIL(VS_ERR_HIP, "trajectory_vote memset: %s", hipGetErrorString(e));
    k_traj_vote<<<dim3((N + 255) / 256), 256, 0, st>>>(idx, labels, F, N, w, spatial_filter, common, winner);
    k_traj_write<<<dim3((unsigned)cdiv64((int64_t)F * N, 256)), 256, 0, st>>>(labels, common, winner, (int64_t)F * N, out);
    VS_CHECK_LAUNCH("trajectory_vote");
    return VS_OK;
}

}  // extern "C"
