// Verifies the operand layout assumed for v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 (e4m3) inputs and unit scales:
// lane l holds row (l & 31) of A (resp. column of B), k = 32*(l >> 5) + 0..31 (32 contiguous bytes = 8 VGPRs);
// C/D as every 32x32 MFMA: col = l & 31, row = (r & 3) + 8*(r >> 2) + 4*(l >> 5).
// build: hipcc --offload-arch=gfx950 -O2 tools/lab/ubench/mx_layout.hip -o tools/lab/ubench/mx_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ unsigned char to_fp8(float f) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(f, 0.f, 0, false);
    return (unsigned char)(w & 0xff);
}

__global__ void k(const float* A, const float* B, float* C) {   // A [32][64], B [32][64] (C = A B^T, 32x32)
    const int l = threadIdx.x, row = l & 31, kb = l >> 5;
    unsigned char a[32], b[32];
    for (int i = 0; i < 32; ++i) {
        a[i] = to_fp8(A[row * 64 + kb * 32 + i]);
        b[i] = to_fp8(B[row * 64 + kb * 32 + i]);
    }
    v8i fa, fb;
    for (int w = 0; w < 8; ++w) {
        fa[w] = a[4 * w] | (a[4 * w + 1] << 8) | (a[4 * w + 2] << 16) | (a[4 * w + 3] << 24);
        fb[w] = b[4 * w] | (b[4 * w + 1] << 8) | (b[4 * w + 2] << 16) | (b[4 * w + 3] << 24);
    }
    v16f c = {0};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * kb) * 32 + row] = c[r];
}

int main() {
    float hA[32 * 64], hB[32 * 64], hC[32 * 32];
    srand(1);
    for (int i = 0; i < 32 * 64; ++i) { hA[i] = (float)(rand() % 9 - 4); hB[i] = (float)(rand() % 7 - 3) * 0.5f; }
    float *dA, *dB, *dC;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dC);
    hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            float ref = 0;
            for (int kk = 0; kk < 64; ++kk) ref += hA[i * 64 + kk] * hB[j * 64 + kk];
            if (ref != hC[i * 32 + j]) { if (bad < 5) printf("C[%d][%d] = %g ref %g\n", i, j, hC[i * 32 + j], ref); ++bad; }
        }
    printf("mx layout check: %d mismatches of 1024\n", bad);
    return bad != 0;
}
