// What does ds_read_b64_tr_b16 return?  LDS element e (16-bit) holds the value e; lane l reads at byte
// address A(l) (two patterns) and prints its four 16-bit results.
// hipcc -O2 --offload-arch=gfx950 tr_read_probe.hip -o tr_read_probe && ./tr_read_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(uint16_t* out, int mode) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr;
    if (mode == 0) addr = l * 8;                                  // consecutive 8-byte chunks
    else addr = ((l & 15) >> 2) * 512 + (l & 3) * 8 + (l >> 4) * 2048;  // 16-lane group g: 4 rows (stride 512 B) x 4 chunks, group base 2048 B
    addr += (unsigned)(uintptr_t)lds;
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = v[0] & 0xffff; out[l * 4 + 1] = v[0] >> 16;
    out[l * 4 + 2] = v[1] & 0xffff; out[l * 4 + 3] = v[1] >> 16;
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (values are LDS element indices = byte address / 2)\n", mode);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d: %5d %5d %5d %5d", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
            if (l % 2 == 1) printf("\n");
        }
    }
    return 0;
}
