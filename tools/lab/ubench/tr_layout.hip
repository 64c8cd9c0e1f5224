// micro-benchmark: which LDS elements does ds_read_b64_tr_b16 hand to which lane?  hipcc --offload-arch=gfx950 -O3
// Every lane passes its own 8-byte-aligned address; LDS holds s[i] = i.  Prints, per lane, the address it passed (in 16-bit
// elements) and the four elements it received, then checks the hypothesis used by k_attention2's V reads:
//   within each 16-lane group, lane i receives element (i & 3) of what lanes 4*j + (i >> 2), j = 0..3, addressed.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const int* offs, short* out) {
    __shared__ __attribute__((aligned(16))) short s[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) s[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(s + offs[l]));
    *reinterpret_cast<s16x4*>(out + l * 4) = r;
}
int main() {
    int h_off[64];
    short h_out[256];
    for (int l = 0; l < 64; ++l) h_off[l] = 1000 * (l >> 4) + ((l & 15) >> 2) * 96 + 4 * (l & 3);   // 4 rows x 16 cols, row stride 96
    int* d_off;
    short* d_out;
    hipMalloc(&d_off, sizeof(h_off));
    hipMalloc(&d_out, sizeof(h_out));
    hipMemcpy(d_off, h_off, sizeof(h_off), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d_off, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d addr %5d -> %5d %5d %5d %5d\n", l, h_off[l], h_out[l * 4], h_out[l * 4 + 1], h_out[l * 4 + 2], h_out[l * 4 + 3]);
        const int g = l & ~15, i = l & 15;
        for (int j = 0; j < 4; ++j) bad += h_out[l * 4 + j] != h_off[g + 4 * j + (i >> 2)] + (i & 3);
    }
    printf("hypothesis result[i][j] = read[4j + (i>>2)][i&3]: %s (%d mismatches)\n", bad ? "WRONG" : "holds", bad);
    return 0;
}
