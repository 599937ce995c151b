// Cycle stamps of the phases of diag_blk5_kernel (the 128x128 leaf of the Cholesky chain) on one random SPD block.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DD5_STAMPS -I ../../gptq-gguf-toolkit_amd/csrc diag5_stamps.hip -o diag5_stamps
#include "../../gptq-gguf-toolkit_amd/csrc/gq_cholesky.hip"
namespace gq { thread_local char g_err[512]; unsigned g_prof_mask = 0; void prof_begin(int, hipStream_t) {} void prof_end(int, hipStream_t) {} }
#include <vector>
int main() {
    using namespace gq;
    const int n = 128;
    std::vector<float> G(n * n), A(n * n);
    srand(1);
    for (auto& g : G) g = rand() / (float)RAND_MAX - 0.5f;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0;
            for (int k = 0; k < n; ++k) s += (double)G[i * n + k] * G[j * n + k];
            A[i * n + j] = (float)(s / n + (i == j ? 0.5 : 0.0));
        }
    float *dA, *dX; int* dF;
    hipMalloc(&dA, n * n * 4); hipMalloc(&dX, n * n * 4); hipMalloc(&dF, 4);
    hipFuncSetAttribute((const void*)diag_blk5_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DIAG5_LDS);
    hipFuncSetAttribute((const void*)diag_blk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DIAG_BLK_LDS);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemcpy(dA, A.data(), n * n * 4, hipMemcpyHostToDevice);
        hipMemset(dF, 0, 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(diag_blk5_kernel, dim3(1), dim3(320), DIAG5_LDS, 0, dA, n, dX, n, dF);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long st[128];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(d5_stamps), sizeof(st));
        printf("rep %d: v5 %.1f us; cycles since start: load %lld |", rep, ms * 1e3, st[1] - st[0]);
        for (int sb = 0; sb < 4; ++sb) {
            printf(" sb%d regs->%lld upd->%lld (waves done:", sb, st[2 + 2 * sb] - st[0], st[3 + 2 * sb] - st[0]);
            for (int w = 0; w < 5; ++w) printf(" %lld", st[16 + 8 * sb + w] - st[0]);
            printf(") |");
        }
        printf(" assembly->%lld store->%lld\n", st[10] - st[0], st[11] - st[0]);
        hipMemcpy(dA, A.data(), n * n * 4, hipMemcpyHostToDevice);
        hipEventRecord(e0);
        hipLaunchKernelGGL(diag_blk_kernel, dim3(1), dim3(256), DIAG_BLK_LDS, 0, dA, n, dX, n, dF);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("        v1 (r02 kernel) %.1f us\n", ms * 1e3);
    }
    return 0;
}
