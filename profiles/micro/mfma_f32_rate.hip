// Sustained v_mfma_f32_32x32x2_f32 rate (the trailing-update / Cholesky fp32 instruction), register-resident
// N(0,1) operands, 1 or 2 waves per SIMD, 2 or 4 independent accumulators per wave.
// hipcc -O3 --offload-arch=gfx950 mfma_f32_rate.hip -o mfma_f32_rate && ./mfma_f32_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(const float* __restrict__ in, float* out, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(t * 16 + i) & 0xffff]; b[i] = in[(t * 16 + 8 + i) & 0xffff]; }
    f32x16 c[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) c[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(s + i) & 7], b[s], c[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += c[i][e];
    out[t] = s;
}

template <int NACC, int WAVES>
void run(const float* in, float* out, const char* name) {
    const int iters = 20000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NACC, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * WAVES * iters * 8 * NACC * 4096.0;
        printf("%s: %d acc/wave, %d waves/CU, rep %d: %.2f ms  %.1f TFLOP/s\n", name, NACC, WAVES, rep, ms, flops / ms / 1e9);
    }
}

int main() {
    const int n = 1 << 16;
    float* h = (float*)malloc(n * sizeof(float));
    srand(1);
    for (int i = 0; i < n; ++i) {
        float u = 0; for (int j = 0; j < 4; ++j) u += rand() / (float)RAND_MAX - 0.5f;
        h[i] = u * 1.7f;
    }
    float *d, *o;
    hipMalloc(&d, n * sizeof(float)); hipMalloc(&o, 256 * 8 * 64 * 4);
    hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice);
    run<2, 8>(d, o, "random");
    run<4, 8>(d, o, "random");
    run<4, 4>(d, o, "random");
    run<2, 4>(d, o, "random");
    hipMemset(d, 0, n * sizeof(float));
    run<4, 8>(d, o, "ZERO data");
    return 0;
}
