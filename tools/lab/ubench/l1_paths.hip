// micro-benchmark: per-CU throughput of 16-byte-per-lane loads from an L2-resident buffer
//   (a) buffer_load_dwordx4 -> VGPR      (b) buffer_load_dwordx4 ... lds (LDS-DMA)
// with 1 / 2 / 4 blocks of 256 threads per CU.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int MODE>
__global__ void __launch_bounds__(256) k(const unsigned* __restrict__ src, unsigned bytes, int iters, unsigned* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, (int)bytes, 0x00020000);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned off = ((blockIdx.x * 4 + wave) * 4096u + lane * 16u) % bytes;
    u32x4 acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {
                u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
                acc += v;
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + wave * 8192 + u * 1024), 16, (int)off, 0, 0, 0);
            }
            off += 1024u * 1024u + 1024u;
            if (off >= bytes) off -= bytes;
        }
        if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (MODE == 1) acc[0] = *reinterpret_cast<unsigned*>(smem + threadIdx.x * 4);
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
    const unsigned bytes = 16u << 20;                      // 16 MiB: L2/MALL resident
    unsigned *src, *out;
    hipMalloc(&src, bytes);
    hipMemset(src, 1, bytes);
    hipMalloc(&out, 1 << 22);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000;
    for (int mode = 0; mode < 2; ++mode)
        for (int bpc : {1, 2, 4}) {
            const int blocks = 256 * bpc;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) k<0><<<blocks, 256, 32768>>>(src, bytes, iters, out);
                else k<1><<<blocks, 256, 32768>>>(src, bytes, iters, out);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double total = (double)blocks * 4 * iters * 8 * 1024.0;
                if (rep) printf("%s blocks/CU %d: %.3f ms  %.2f TB/s  %.1f B/clk/CU @2.4GHz\n", mode ? "lds-dma " : "to-vgpr ", bpc, ms,
                                total / ms / 1e9, total / 256 / (ms * 1e-3 * 2.4e9));
            }
        }
    return 0;
}
