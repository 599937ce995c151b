// r05: what does the L2 -> LDS operand path deliver with the SYRK's access pattern, and what changes it?
// The four-wave SYRK (syrk16_256w_kernel) with its MFMAs and fragment reads REMOVED runs as long as the whole kernel
// (profiles/r05_syrk_decompose.txt): the kernel is bound by operand delivery at ~9.8 TB/s.  This standalone kernel is that
// delivery stream alone -- 256 workgroups x 4 waves, per k32 half-stage every wave brings 4 + 4 pieces of 1 KiB (two 512-byte
// row segments each) of the A and B panels of its 256 x 256 tile, ring pacing by s_waitcnt vmcnt(N) + s_barrier -- with knobs:
//   mode 0  LDS-DMA ring as shipped (vmcnt(8) + barrier per step)        mode 1  no barrier
//   mode 2  three half-stages in flight (vmcnt(16))                         mode 3  loads into VGPRs instead of LDS-DMA
//   share 0 the 4 x 8 super-tile per XCD (12 panels per 32 workgroups), 1 every workgroup of an XCD the same two panels,
//         2 every workgroup its own two panels
//   pad     extra bytes per token row (0: rows 2 C bytes apart, a multiple of 4 KiB for C = 14336 and 4096)
// hipcc -O3 -std=c++17 --offload-arch=gfx950 dma_stream.hip -o dma_stream ; ./dma_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define WDL(vo, base, ldsa) \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vo), "s"(base), "s"(ldsa) : "memory")

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void stream_kernel(
    const char* __restrict__ X, int64_t ld, int nhs, int share, int rounds, unsigned* sink, int swz) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    unsigned acc = 0;
    for (int rd = 0; rd < rounds; ++rd) {
        int ti, tj;
        if (share == 0) { ti = 4 * ((xcd + rd) % 14) + (slot >> 3); tj = 8 * ((xcd * 3 + rd) % 7) + (slot & 7); }
        else if (share == 1) { ti = (xcd + rd) % 56; tj = (xcd + 8 + rd) % 56; }
        else { ti = ((int)blockIdx.x + rd * 7) % 56; tj = ((int)blockIdx.x * 5 + 3 + rd) % 56; }
        const int rb = 8 * wid;
        unsigned voff[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r_ = rb + 2 * u + (lane >> 5), s16 = lane & 31;
            const int g_ = (r_ & 3) | (((r_ >> 3) & 1) << 2);
            unsigned in_row = (unsigned)(s16 << 4);                                       // swz 0: lanes in address order
            if (swz == 1) in_row = (unsigned)((((s16 >> 1) ^ g_) << 5) + ((s16 & 1) << 4));  // the shipped kernel: 32-byte slots permuted
            if (swz == 2) in_row = (unsigned)((((s16 >> 2) ^ (g_ & 3)) << 6) + ((s16 & 3) << 4));  // 64-byte units permuted
            if (swz == 3) in_row = (unsigned)((((s16 >> 3) ^ (g_ & 1)) << 7) + ((s16 & 7) << 4));  // 128-byte units permuted
            voff[u] = (unsigned)(r_ * ld) + in_row;
        }
        const unsigned ldsw = lds0 + (unsigned)(rb * 512);
        const int64_t hstride = 32 * ld;
        const char* gA = X + (int64_t)ti * 512;
        const char* gB = X + (int64_t)tj * 512;
        __syncthreads();
        for (int n = 0; n < nhs; ++n) {
            const char* a = gA + (int64_t)n * hstride;
            const char* b = gB + (int64_t)n * hstride;
            const unsigned sl = (unsigned)((n & 3) * 32768);
            if constexpr (MODE == 3) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    uint4 va, vb;
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(va) : "v"(voff[u]), "s"(a) : "memory");
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(vb) : "v"(voff[u]), "s"(b) : "memory");
                    asm volatile("s_waitcnt vmcnt(14)" ::: "memory");  // registers are reused: keep two half-stages' worth in flight at most
                    acc ^= va.x ^ vb.x;
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) WDL(voff[u], a, ldsw + sl + (unsigned)(u * 1024));
#pragma unroll
                for (int u = 0; u < 4; ++u) WDL(voff[u], b, ldsw + sl + 16384u + (unsigned)(u * 1024));
                if constexpr (MODE == 0) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
                if constexpr (MODE == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                if constexpr (MODE == 2) asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (acc == 0x12345u) sink[0] = acc;
}

template <int MODE>
static double run(const char* X, int64_t ld, int nhs, int share, int rounds, unsigned* sink, int swz = 0) {
    hipFuncSetAttribute((const void*)stream_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 6; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(stream_kernel<MODE>, dim3(256), dim3(256), 131072, 0, X, ld, nhs, share, rounds, sink, swz);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r >= 2 && ms < best) best = ms;
    }
    const double bytes = 256.0 * rounds * nhs * 32768.0;
    return bytes / (best * 1e-3) / 1e12;
}

int main(int argc, char** argv) {
    const int64_t T = 65536, C = 14336;
    const int nhs = (int)(T / 32), rounds = 3;
    unsigned* sink; hipMalloc(&sink, 64);
    printf("DMA-only operand stream, 256 workgroups x 4 waves, %d rounds of %d half-stages (32 KiB each): TB/s delivered  [x 128 flop/B = the PFLOP/s it would feed]\n", rounds, nhs);
    {
        const int64_t ld = C * 2;
        char* X; hipMalloc(&X, (size_t)(T + 64) * ld);
        hipMemset(X, 0, (size_t)(T + 64) * ld);
        for (int swz = 0; swz < 4; ++swz)
            printf("source order within a 512-byte row segment: %s | ring as shipped %6.2f TB/s | no barrier %6.2f | VGPR loads %6.2f\n",
                   swz == 0 ? "ascending          " : swz == 1 ? "32-B slots permuted (shipped)" : swz == 2 ? "64-B units permuted" : "128-B units permuted",
                   run<0>(X, ld, nhs, 0, rounds, sink, swz), run<1>(X, ld, nhs, 0, rounds, sink, swz), run<3>(X, ld, nhs, 0, rounds, sink, swz));
        hipFree(X);
    }
    for (int64_t pad : {0LL, 256LL}) {
        const int64_t ld = C * 2 + pad;
        char* X; hipMalloc(&X, (size_t)(T + 64) * ld);
        hipMemset(X, 0, (size_t)(T + 64) * ld);
        for (int share = 0; share < 3; ++share) {
            const double a = run<0>(X, ld, nhs, share, rounds, sink), b = run<1>(X, ld, nhs, share, rounds, sink),
                         c = run<2>(X, ld, nhs, share, rounds, sink), d = run<3>(X, ld, nhs, share, rounds, sink);
            printf("pad %5lld share %d | ring as shipped %6.2f (%.2f PF) | no barrier %6.2f | 3 half-stages in flight %6.2f | VGPR loads %6.2f\n",
                   (long long)pad, share, a, a * 128e-3, b, c, d);
        }
        hipFree(X);
    }
    return 0;
}
