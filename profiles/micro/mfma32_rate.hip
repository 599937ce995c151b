// What does v_mfma_f32_32x32x2_f32 sustain?  Register-resident loop, NACC accumulators per wave used round-robin,
// W waves per workgroup (one workgroup per CU): TFLOP/s and the implied cycles per instruction per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 profiles/micro/mfma32_rate.hip -o /tmp/mfma32_rate && /tmp/mfma32_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = a0 + threadIdx.x * 1e-7f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int waves, float* out) {
    const int iters = 4000, cus = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<NACC>, dim3(cus), dim3(64 * waves), 0, 0, out, iters, 1.0f, 1e-3f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)iters * 16 * NACC * waves * cus;  // instructions
        if (rep) printf("acc/wave %d  waves/CU %2d: %7.1f TFLOP/s  (%.1f ns per instruction per SIMD)\n", NACC, waves,
                        n * 4096 / ms / 1e9, ms * 1e6 / (n / (cus * 4.0)));
    }
}
int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * 4);
    for (int w : {4, 8, 16}) { run<1>(w, out); run<2>(w, out); run<4>(w, out); }
    return 0;
}
