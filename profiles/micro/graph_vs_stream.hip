// Would hipGraph shorten a chain of small dependent launches (gq_h_prepare: ~560, a column loop: ~300)?
// N dependent launches of a trivial kernel on one stream: (a) plain launches, host far ahead of the device (a 3 ms spin kernel in
// front, so the GPU-side gap is what is timed), (b) the same N launches captured once into a hipGraph and replayed.
// hipcc -O3 --offload-arch=gfx950 graph_vs_stream.hip -o graph_vs_stream && ./graph_vs_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void k0(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0f; }
__global__ void k64(float* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.0f; }
__global__ void spin(float* p, long long cycles) {
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (threadIdx.x == 0) p[1] += 1.0f;
}
int main() {
    float* d; (void)hipMalloc(&d, 4096); (void)hipMemset(d, 0, 4096);
    hipStream_t st; (void)hipStreamCreate(&st);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int N = 600;
    for (int wg = 0; wg < 2; ++wg) {
        for (int rep = 0; rep < 3; ++rep) {
            // (a) stream launches behind a spin kernel: the host finishes enqueueing while the spin runs
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, d, 6000000LL);
            (void)hipEventRecord(e0, st);
            auto h0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) {
                if (wg) hipLaunchKernelGGL(k64, dim3(64), dim3(256), 0, st, d); else hipLaunchKernelGGL(k0, dim3(1), dim3(64), 0, st, d);
            }
            auto h1 = std::chrono::steady_clock::now();
            (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("%s, stream : %.2f us per launch on the device, %.2f us per launch on the host\n", wg ? "64 workgroups" : "1 workgroup  ",
                                 1e3 * ms / N, std::chrono::duration<double, std::micro>(h1 - h0).count() / N);
        }
        hipGraph_t g; hipGraphExec_t ge;
        (void)hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int i = 0; i < N; ++i) {
            if (wg) hipLaunchKernelGGL(k64, dim3(64), dim3(256), 0, st, d); else hipLaunchKernelGGL(k0, dim3(1), dim3(64), 0, st, d);
        }
        (void)hipStreamEndCapture(st, &g);
        (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, d, 6000000LL);
            (void)hipEventRecord(e0, st);
            auto h0 = std::chrono::steady_clock::now();
            (void)hipGraphLaunch(ge, st);
            auto h1 = std::chrono::steady_clock::now();
            (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("%s, graph  : %.2f us per launch on the device, %.2f us per launch on the host\n", wg ? "64 workgroups" : "1 workgroup  ",
                                 1e3 * ms / N, std::chrono::duration<double, std::micro>(h1 - h0).count() / N);
        }
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    }
    return 0;
}
