// micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 (1, 2 waves per SIMD) -- hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) double f64x4;
__global__ void k(double* out, int iters, double a, double b) {
    f64x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
    double* d;
    hipMalloc(&d, 1 << 24);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int threads : {256, 512}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            k<<<256, threads>>>(d, iters, 1.0, 2.0);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double mf = 4.0 * iters;                       // MFMAs per wave
            const double waves_per_simd = threads / 256.0;
            printf("threads %d: %.3f ms, %.1f ns per MFMA per SIMD (=%.0f clk @2.4GHz), %.1f TF/s f64\n", threads, ms,
                   ms * 1e6 / (mf * waves_per_simd), ms * 1e6 / (mf * waves_per_simd) * 2.4,
                   256.0 * (threads / 64) * mf * 2048 / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
