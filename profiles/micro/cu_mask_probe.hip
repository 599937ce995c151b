// Which CUs does a stream created with hipExtStreamCreateWithCUMask run on?  One workgroup per slot records its
// XCC id and HW_ID (SE / CU); the host prints the histogram for a few masks.  hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <map>
__global__ void where(unsigned* out, int spin) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = xcc & 0xf;
        out[2 * blockIdx.x + 1] = hw;
    }
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
}
static void run(hipStream_t st, const char* tag) {
    const int n = 4096;
    unsigned* d;
    hipMalloc(&d, 2 * n * sizeof(unsigned));
    hipLaunchKernelGGL(where, dim3(n), dim3(256), 0, st, d, 200);
    hipStreamSynchronize(st);
    std::vector<unsigned> h(2 * n);
    hipMemcpy(h.data(), d, 2 * n * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<unsigned, std::map<unsigned, int>> cnt;  // xcc -> (se, cu) -> count
    for (int i = 0; i < n; ++i) cnt[h[2 * i]][(h[2 * i + 1] >> 8) & 0xff]++;  // HW_ID bits 8..: cu_id(4) sh_id(1) se_id(3)
    int cus = 0;
    printf("%s:", tag);
    for (auto& x : cnt) {
        printf(" xcc%u:%zu", x.first, x.second.size());
        cus += (int)x.second.size();
    }
    printf("  -> %d distinct (xcc, se/cu) slots\n", cus);
    hipFree(d);
}
int main() {
    hipStream_t s0;
    hipStreamCreate(&s0);
    run(s0, "plain stream");
    for (int words = 1; words <= 8; words *= 2) {
        std::vector<uint32_t> mask(8, 0);
        for (int w = 0; w < words; ++w) mask[w] = 0xffffffffu;
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask.data());
        if (e != hipSuccess) { printf("mask %d words: %s\n", words, hipGetErrorString(e)); continue; }
        char tag[64];
        snprintf(tag, sizeof tag, "first %d bits set", 32 * words);
        run(s, tag);
    }
    {   // every fourth bit
        std::vector<uint32_t> mask(8, 0x11111111u);
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, 8, mask.data()) == hipSuccess) run(s, "every 4th bit");
    }
    {   // 3 of every 4 bits
        std::vector<uint32_t> mask(8, 0x77777777u);
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, 8, mask.data()) == hipSuccess) run(s, "3 of 4 bits");
    }
    return 0;
}
