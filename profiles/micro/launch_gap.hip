// Dependent-launch cost on one stream: N back-to-back launches of trivial kernels (same kernel / two kernels
// alternating / with 100 KB of dynamic LDS), wall time per launch.
// hipcc -O3 --offload-arch=gfx950 launch_gap.hip -o launch_gap && ./launch_gap
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k0(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0f; }
__global__ void k1(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[1] += 2.0f; }
__global__ void klds(float* p) {
    extern __shared__ float s[];
    s[threadIdx.x] = p[0];
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) p[2] += s[1];
}
int main() {
    float* d; (void)hipMalloc(&d, 1024); (void)hipMemset(d, 0, 1024);
    (void)hipFuncSetAttribute((const void*)klds, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipStream_t st; (void)hipStreamCreate(&st);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int N = 2000;
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0, st);
            for (int i = 0; i < N; ++i) {
                if (mode == 0) hipLaunchKernelGGL(k0, dim3(1), dim3(64), 0, st, d);
                else if (mode == 1) { if (i & 1) hipLaunchKernelGGL(k1, dim3(1), dim3(64), 0, st, d); else hipLaunchKernelGGL(k0, dim3(1), dim3(64), 0, st, d); }
                else if (mode == 2) hipLaunchKernelGGL(klds, dim3(1), dim3(256), 100 * 1024, st, d);
                else hipLaunchKernelGGL(k0, dim3(1024), dim3(256), 0, st, d);
            }
            (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("mode %d (%s): %.2f us per launch\n", mode,
                            mode == 0 ? "same trivial kernel" : mode == 1 ? "two kernels alternating" : mode == 2 ? "100 KB dynamic LDS" : "1024 workgroups",
                            1e3 * ms / N);
        }
    }
    return 0;
}
