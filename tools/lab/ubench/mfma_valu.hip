// micro-benchmark: what does VALU work cost beside v_mfma_f32_32x32x16_f16 on one SIMD?  hipcc --offload-arch=gfx950 -O3
// One loop iteration = 4 MFMAs on 4 independent accumulators, each followed by N filler instructions of one kind on independent
// registers.  Reported per MFMA, in units of the bare MFMA loop (= 32 cycles): 1.00 means the fillers were free.
// Run with 1 wave per SIMD (256 threads / CU) and 2 (512); a third mode puts MFMAs in waves 0-3 and fillers in waves 4-7.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int N>
__device__ __forceinline__ void fillers(float (&x)[8], f32x2 (&y)[8], unsigned (&w)[8], unsigned& sreg) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
        if (KIND == 2) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x[i]));
        if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(y[i]));
        if (KIND == 4) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(w[i]) : "v"(x[i]));
        if (KIND == 5) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(x[i]));
        if (KIND == 6) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(y[i]));
        if (KIND == 7) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i]));
        if (KIND == 8) asm volatile("v_add_u32 %0, %0, %0" : "+v"(w[i]));
        if (KIND == 9) asm volatile("s_add_u32 s20, s20, 1" ::: "s20", "scc");
        if (KIND == 10) asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(w[i]));
    }
}

// MODE 0: every wave runs MFMA + fillers; MODE 1: waves 0-3 MFMAs only, waves 4-7 the same number of fillers only
template <int KIND, int N, int MODE>
__global__ void __launch_bounds__(512) k(float* out, int iters, f16x8 a, f16x8 b) {
    f32x16 c[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    float x[8];
    f32x2 y[8];
    unsigned w[8];
    unsigned sreg = 1;
    for (int i = 0; i < 8; ++i) { x[i] = -1.0f - i; y[i] = f32x2{1.f, 2.f}; w[i] = 0; }
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a), "v"(b));
                fillers<KIND, N>(x, y, w, sreg);
            }
        }
    } else if (__builtin_amdgcn_readfirstlane(threadIdx.x) < 256) {      // one uniform branch, then two branch-free loops
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a), "v"(b));
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) fillers<KIND, N>(x, y, w, sreg);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][5];
    for (int i = 0; i < 8; ++i) s += x[i] + y[i][0] + (float)w[i];
    s += (float)sreg;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int N, int MODE>
float run(float* d, int threads, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.5f; b[i] = (_Float16)0.25f; }
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<KIND, N, MODE><<<256, threads>>>(d, iters, a, b);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}

template <int KIND>
void kind(float* d, const char* name, int iters) {
    const float base1 = run<0, 0, 0>(d, 256, iters), base2 = run<0, 0, 0>(d, 512, iters);
    printf("%-14s 1 wave/SIMD: N=2 %.2f  N=4 %.2f  N=6 %.2f  N=8 %.2f | 2 waves/SIMD (per MFMA of either wave): N=2 %.2f N=4 %.2f N=6 %.2f N=8 %.2f | split waves (4 MFMA waves + 4 filler waves): N=4 %.2f N=8 %.2f\n",
           name, run<KIND, 2, 0>(d, 256, iters) / base1, run<KIND, 4, 0>(d, 256, iters) / base1, run<KIND, 6, 0>(d, 256, iters) / base1,
           run<KIND, 8, 0>(d, 256, iters) / base1, run<KIND, 2, 0>(d, 512, iters) / base2, run<KIND, 4, 0>(d, 512, iters) / base2,
           run<KIND, 6, 0>(d, 512, iters) / base2, run<KIND, 8, 0>(d, 512, iters) / base2, run<KIND, 4, 1>(d, 512, iters) / base1,
           run<KIND, 8, 1>(d, 512, iters) / base1);
}

template <int KIND>
void split_small(float* d, const char* name, int iters) {
    const float base1 = run<0, 0, 0>(d, 256, iters);
    printf("%-14s split waves (4 MFMA waves + 4 filler waves), fillers per MFMA slot: N=1 %.2f  N=2 %.2f  N=4 %.2f\n", name,
           run<KIND, 1, 1>(d, 512, iters) / base1, run<KIND, 2, 1>(d, 512, iters) / base1, run<KIND, 4, 1>(d, 512, iters) / base1);
}

int main() {
    float* d;
    hipMalloc(&d, 1 << 22);
    const int iters = 20000;
    const float b1 = run<0, 0, 0>(d, 256, iters), b2 = run<0, 0, 0>(d, 512, iters);
    printf("bare MFMA loop: 1 wave/SIMD %.3f ms (%.1f ns per MFMA), 2 waves/SIMD %.3f ms (%.1f ns per MFMA per SIMD)\n", b1,
           b1 * 1e6 / (4.0 * iters), b2, b2 * 1e6 / (8.0 * iters));
    kind<1>(d, "v_exp_f32", iters);
    kind<2>(d, "v_add_f32", iters);
    kind<3>(d, "v_pk_add_f32", iters);
    kind<4>(d, "v_cvt_pk_f16", iters);
    kind<5>(d, "v_max3_f32", iters);
    kind<6>(d, "v_pk_fma_f32", iters);
    kind<7>(d, "v_fma_f32", iters);
    kind<8>(d, "v_add_u32", iters);
    kind<9>(d, "s_add_u32", iters);
    kind<10>(d, "v_mul_lo_u32", iters);
    split_small<2>(d, "v_add_f32", iters);
    split_small<8>(d, "v_add_u32", iters);
    split_small<9>(d, "s_add_u32", iters);
    split_small<10>(d, "v_mul_lo_u32", iters);
    return 0;
}
