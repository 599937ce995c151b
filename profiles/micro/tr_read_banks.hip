// Bank behaviour of ds_read_b64_tr_b16 for the natural-layout SYRK image ([32 tokens][256 channels], 512-B rows).
// hipcc -O2 --offload-arch=gfx950 tr_read_banks.hip -o tr_read_banks && ./tr_read_banks
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k(unsigned long long* out, int mode, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int l = threadIdx.x & 63, g = l >> 4, j = l & 15;
    unsigned a;
    const int row = 8 * g + (j >> 2);   // first read of a fragment: rows 8kc .. 8kc+3
    if (mode == 0) a = l * 8;                                                        // contiguous
    else if (mode == 1) a = row * 512 + (j & 3) * 8;                                 // natural rows, no swizzle
    else if (mode == 2) a = row * 512 + (((row & 3) | (((row >> 3) & 1) << 2)) * 32) + (j & 3) * 8;  // 32-B XOR swizzle g(r)
    else a = row * 544 + (j & 3) * 8;                                                // rows padded by 32 B
    a += (unsigned)(uintptr_t)lds + (threadIdx.x >> 6) * 20480;
    u32x2 acc = {0, 0};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        u32x2 v0, v1, v2, v3;
        asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:2048\n\t"
                     "ds_read_b64_tr_b16 %2, %4 offset:32\n\tds_read_b64_tr_b16 %3, %4 offset:2080\n\ts_waitcnt lgkmcnt(0)"
                     : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(a) : "memory");
        acc += v0 + v1 + v2 + v3;
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc[0] == 0x12345678u) out[1000] = acc[1];
}
int main() {
    unsigned long long* d; hipMalloc(&d, 8192 * 8);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    const char* names[4] = {"contiguous 512 B", "natural rows (512 B stride)", "32-B XOR swizzle g(r)", "rows padded to 544 B"};
    for (int mode = 0; mode < 4; ++mode) {
        unsigned long long h[4];
        const int iters = 20000;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k, dim3(256), dim3(256), 96 * 1024, 0, d, mode, iters);
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        }
        printf("%-30s %.2f cycles per wave-instruction (4 waves/CU issuing)\n", names[mode], (double)h[0] / iters / 4);
    }
    return 0;
}
