// Cycle stamps of one workgroup of the GPTQ far update (gemm32_chain_full_kernel's tile function, STAMP instantiation): per wave,
// the shader-clock cycles of the chunk loop, of the vmcnt + barrier of every chunk, and of its commit + fetch block -- measured
// inside a launch of the bench shape (M = 4096, N = 8192, K = 1024: 2048 tiles, the stamped workgroup in the middle of the grid).
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off far_stamps.hip -o far_stamps   (run: ./far_stamps)
#include "../../gptq-gguf-toolkit_amd/csrc/gq_gemm32.hpp"
#include <vector>
namespace gq {
thread_local char g_err[512];
unsigned g_prof_mask = 0;
void prof_begin(int, hipStream_t) {}
void prof_end(int, hipStream_t) {}
int64_t opt(Opt) { return 0; }
template <bool BDMA, bool STAMP, int CG>
__global__ __launch_bounds__(512, 2) void far_probe_kernel(float* Cmat, int64_t ldc, const float* A, int64_t lda, const float* B,
                                                          int64_t ldb, int64_t K, unsigned long long* stamps, int px, int py) {
    unsigned long long* st = (STAMP && (int)blockIdx.x == px && (int)blockIdx.y == py) ? stamps : nullptr;
    g32_chain_full_tile<128, BDMA, STAMP, CG>(Cmat, ldc, A, lda, B, ldb, K, (int64_t)blockIdx.y * 128, (int64_t)blockIdx.x * 128, st);
}
}  // namespace gq
template <bool BDMA, bool STAMP, int CG>
static float run(float* C, const float* A, const float* B, int64_t M, int64_t N, int64_t K, unsigned long long* st, int reps) {
    constexpr int LDS = 3 * gq::G32<128>::STAGE_FLOATS * 4;
    hipFuncSetAttribute((const void*)gq::far_probe_kernel<BDMA, STAMP, CG>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    dim3 grid((unsigned)(N / 128), (unsigned)(M / 128));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 30; ++r)  // warm-up: the clock settles
        hipLaunchKernelGGL((gq::far_probe_kernel<BDMA, STAMP, CG>), grid, dim3(512), LDS, 0, C, N, A, K, B, N, K, st, (int)(N / 256), (int)(M / 256));
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((gq::far_probe_kernel<BDMA, STAMP, CG>), grid, dim3(512), LDS, 0, C, N, A, K, B, N, K, st, (int)(N / 256), (int)(M / 256));
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}
template <bool BDMA, int CG>
static void report(const char* tag, float* C, const float* A, const float* B, int64_t M, int64_t N, int64_t K, unsigned long long* dst) {
    const double fl = 2.0 * M * N * K;
    const float plain = run<BDMA, false, CG>(C, A, B, M, N, K, dst, 5);
    const float stamped = run<BDMA, true, CG>(C, A, B, M, N, K, dst, 5);
    unsigned long long h[32];
    hipMemcpy(h, dst, sizeof(h), hipMemcpyDeviceToHost);
    printf("[%s] plain %.3f ms = %.1f TFLOP/s (%.3f of 157.3) | with stamps %.3f ms\n", tag, plain, fl / plain / 1e9, fl / plain / 1e9 / 157.3, stamped);
    printf("  per wave: loop cycles / chunk, of which vmcnt + barrier, commit + fetch block (ideal: 32 MFMAs x 64 cycles x 2 waves per SIMD = 4096)\n");
    for (int w = 0; w < 8; ++w) {
        const double n = (double)h[4 * w + 3];
        printf("  wave %d: %7.0f  barrier %6.0f (%4.1f %%)  commit %6.0f (%4.1f %%)\n", w, h[4 * w] / n, h[4 * w + 1] / n, 100.0 * h[4 * w + 1] / h[4 * w],
               h[4 * w + 2] / n, 100.0 * h[4 * w + 2] / h[4 * w]);
    }
}
int main() {
    const int64_t M = 4096, N = 8192, K = 1024;
    float *A, *B, *C; unsigned long long* st;
    hipMalloc(&A, M * K * 4); hipMalloc(&B, K * N * 4); hipMalloc(&C, M * N * 4); hipMalloc(&st, 32 * 8);
    std::vector<float> h((size_t)M * N);
    srand(1);
    for (auto& x : h) x = rand() / (float)RAND_MAX - 0.5f;
    hipMemcpy(A, h.data(), M * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), K * N * 4, hipMemcpyHostToDevice);
    hipMemcpy(C, h.data(), M * N * 4, hipMemcpyHostToDevice);
    report<false, 3>("B through registers, commit behind group 3 (r03)", C, A, B, M, N, K, st);
    report<true, 3>("B by LDS-DMA, commit behind group 3", C, A, B, M, N, K, st);
    report<true, 1>("B by LDS-DMA, commit behind group 1 (shipped)", C, A, B, M, N, K, st);
    report<true, 5>("B by LDS-DMA, commit behind group 5", C, A, B, M, N, K, st);
    return 0;
}
