// gq_cholesky.hip -- K2 + K3: Hessian preparation (GPTQ.quantization_pre_step's
// dead-channel fix, GPTQ._prepare: zero-column masking, damping) and
//   U = cholesky(cholesky_inverse(cholesky(H)), upper=True)     (gptq.py:319-320)
//
// The three-factorisation chain of the reference (4n^3/3 flop) is replaced by the
// mathematically identical
//   A = J H J  (J = index reversal),  A = M M^T (lower Cholesky),  U = J M^-1 J
// (one potrf + one triangular inverse): H = (JMJ)(JMJ)^T with JMJ upper, hence
// H^-1 = (JMJ)^-T (JMJ)^-1 and U = (JMJ)^-1 is THE upper factor with U^T U = H^-1.
// This stage is tolerance-class in the reference itself (LAPACK vs MAGMA/cuSOLVER).
//
//   potrf : right-looking, nb = 128.  Diagonal block: one workgroup factors it in
//           LDS and also inverts it; panel  A21 <- A21 * L11^-T  and the trailing
//           symmetric update  A22 -= L21 L21^T  run on the fp32 matrix cores.
//   trtri : recursive halving, X21 = -X22 * (M21 * X11), two GEMMs per node.
#include "gq_common.hpp"
#include "gq_gemm32.hpp"

namespace gq {

constexpr int NB = 128;
constexpr int NBP = NB + 1;

// ------------------------------------------------------------------ prelude
// dead[j] = (H[j,j] == 0)                               gptq.py:134
// zc[j]   = dead[j] || all_r(W[r,j] == 0)               gptq.py:308 (after W[:,dead]=0, :141)
__global__ __launch_bounds__(256) void col_flags_kernel(const float* __restrict__ H, const float* __restrict__ W,
                                                        int64_t R, int64_t C, uint8_t* __restrict__ dead,
                                                        uint8_t* __restrict__ zc) {
    const int64_t j = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const int ty = threadIdx.x >> 6;
    __shared__ int nz[4][64];
    int any = 0;
    if (j < C)
        for (int64_t r = ty; r < R; r += 4) any |= (W[r * C + j] != 0.0f);
    nz[ty][threadIdx.x & 63] = any;
    __syncthreads();
    if (ty == 0 && j < C) {
        int a = nz[0][threadIdx.x] | nz[1][threadIdx.x] | nz[2][threadIdx.x] | nz[3][threadIdx.x];
        uint8_t dd = H[j * C + j] == 0.0f;
        dead[j] = dd;
        zc[j] = dd || !a;
    }
}

__global__ __launch_bounds__(256) void zero_dead_cols_kernel(float* __restrict__ W, int64_t R, int64_t C,
                                                             const uint8_t* __restrict__ dead) {
    const int64_t total = R * C;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x)
        if (dead[t % C]) W[t] = 0.0f;  // gptq.py:141
}

// H[zc,:] = 0; H[:,zc] = 0; H[zc,zc] = 1  (gptq.py:311-313; also covers H[dead,dead]=1, :135)
__global__ __launch_bounds__(256) void mask_h_kernel(float* __restrict__ H, int64_t C, const uint8_t* __restrict__ zc) {
    const int64_t total = C * C;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t / C, j = t % C;
        if (zc[i] || zc[j]) H[t] = (i == j) ? 1.0f : 0.0f;
    }
}

// damp = rel_damp * mean(diag H); H_ii += damp  (gptq.py:315-316).  One workgroup.
__global__ __launch_bounds__(1024) void damp_kernel(float* __restrict__ H, int64_t C, float rel_damp) {
    __shared__ double part[1024];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < C; i += 1024) acc += (double)H[i * C + i];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    const float damp = rel_damp * (float)(part[0] / (double)C);
    for (int64_t i = threadIdx.x; i < C; i += 1024) H[i * C + i] += damp;
}

// A[i,j] = H[n-1-i, n-1-j]
__global__ __launch_bounds__(256) void reverse_copy_kernel(float* __restrict__ A, const float* __restrict__ H,
                                                           int64_t n) {
    const int64_t total = n * n;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x)
        A[t] = H[total - 1 - t];
}

// U[i,j] = X[n-1-i, n-1-j] for j >= i, 0 below; identity if *flag
__global__ __launch_bounds__(256) void finish_u_kernel(float* __restrict__ U, const float* __restrict__ X, int64_t n,
                                                       const int* __restrict__ flag) {
    const int64_t total = n * n;
    const bool bad = *flag != 0;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t / n, j = t % n;
        float v;
        if (bad) v = (i == j) ? 1.0f : 0.0f;  // gptq.py:321-323
        else v = (j >= i) ? X[total - 1 - t] : 0.0f;
        U[t] = v;
    }
}

// --------------------------------------------------- diagonal block: potrf + inverse
// One workgroup (256 threads).  A_kk (lower) -> L_kk in place; Dinv = L_kk^-1 (dense
// 128x128 with zeros above the diagonal).  A non-positive pivot raises *flag.
__global__ __launch_bounds__(256) void diag_potrf_inv_kernel(float* __restrict__ A, int64_t lda,
                                                             float* __restrict__ Dinv, int* __restrict__ flag) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* S = smem;             // [NB][NBP]
    float* Xs = smem + NB * NBP; // [NB][NBP]
    const int tid = threadIdx.x;
    for (int idx = tid; idx < NB * NB; idx += 256) {
        int i = idx / NB, j = idx % NB;
        S[i * NBP + j] = (j <= i) ? A[i * lda + j] : 0.0f;
        Xs[i * NBP + j] = 0.0f;
    }
    const int ti = tid >> 4, tc = tid & 15;
    for (int j = 0; j < NB; ++j) {
        __syncthreads();
        if (tid == 0) {
            float piv = S[j * NBP + j];
            if (!(piv > 0.0f)) {  // also catches NaN
                *flag = 1;
                piv = 1.0f;
            }
            S[j * NBP + j] = sqrtf(piv);
        }
        __syncthreads();
        const float ljj = S[j * NBP + j];
        for (int i = j + 1 + tid; i < NB; i += 256) S[i * NBP + j] = S[i * NBP + j] / ljj;
        __syncthreads();
        for (int i = j + 1 + ti; i < NB; i += 16) {
            const float lij = S[i * NBP + j];
            for (int c = j + 1 + tc; c <= i; c += 16) S[i * NBP + c] = fmaf(-lij, S[c * NBP + j], S[i * NBP + c]);
        }
    }
    __syncthreads();
    // inverse by forward substitution, one thread per column
    if (tid < NB) {
        const int c = tid;
        Xs[c * NBP + c] = 1.0f / S[c * NBP + c];
        for (int i = c + 1; i < NB; ++i) {
            float acc = 0.0f;
            for (int p = c; p < i; ++p) acc = fmaf(S[i * NBP + p], Xs[p * NBP + c], acc);
            Xs[i * NBP + c] = -acc / S[i * NBP + i];
        }
    }
    __syncthreads();
    for (int idx = tid; idx < NB * NB; idx += 256) {
        int i = idx / NB, j = idx % NB;
        if (j <= i) A[i * lda + j] = S[i * NBP + j];
        Dinv[idx] = Xs[i * NBP + j];
    }
}

__global__ __launch_bounds__(256) void copy_block_kernel(float* __restrict__ dst, int64_t ldd,
                                                         const float* __restrict__ src) {
    for (int idx = threadIdx.x; idx < NB * NB; idx += 256) dst[(idx / NB) * ldd + (idx % NB)] = src[idx];
}

size_t h_prepare_workspace_bytes(int64_t R, int64_t C) {
    (void)R;
    const size_t n2 = (size_t)C * (size_t)C * sizeof(float);
    return 2 * n2 + (size_t)C * NB * sizeof(float) + 2 * (size_t)C + 1024;
}

// X[lo:hi, lo:hi] = inverse of the lower-triangular M[lo:hi, lo:hi] (block indices)
static int trtri_rec(const float* M, float* X, float* Tmp, const float* Dinv, int64_t n, int64_t lo, int64_t hi,
                     hipStream_t st) {
    if (hi - lo == 1) {
        hipLaunchKernelGGL(copy_block_kernel, dim3(1), dim3(256), 0, st, X + (lo * NB) * n + lo * NB, n,
                           Dinv + lo * NB * NB);
        GQ_LAUNCH_CHECK();
        return GQ_OK;
    }
    const int64_t mid = (lo + hi) / 2;
    int rc;
    if ((rc = trtri_rec(M, X, Tmp, Dinv, n, lo, mid, st))) return rc;
    if ((rc = trtri_rec(M, X, Tmp, Dinv, n, mid, hi, st))) return rc;
    const int64_t m2 = (hi - mid) * NB, m1 = (mid - lo) * NB;
    const int64_t o21 = (mid * NB) * n + lo * NB, o11 = (lo * NB) * n + lo * NB, o22 = (mid * NB) * n + mid * NB;
    // T = M21 * X11 ;  X21 = -(X22 * T)
    if ((rc = launch_gemm32<false, 1, false>(Tmp + o21, n, M + o21, n, X + o11, n, m2, m1, m1, st))) return rc;
    return launch_gemm32<false, 2, false>(X + o21, n, X + o22, n, Tmp + o21, n, m2, m1, m2, st);
}

int h_prepare(float* H, float* W, int64_t R, int64_t C, float rel_damp, float* U, int* not_invertible, void* ws,
              size_t ws_bytes, hipStream_t st) {
    if (!H || !W || !U || !not_invertible) GQ_FAIL(GQ_E_NULL, "gq_h_prepare: null pointer");
    if (R <= 0 || C <= 0 || C % NB) GQ_FAIL(GQ_E_BAD_SHAPE, "gq_h_prepare: R=%ld C=%ld (C %% 128 != 0)", (long)R, (long)C);
    const size_t need = h_prepare_workspace_bytes(R, C);
    if (!ws || ws_bytes < need) GQ_FAIL(GQ_E_WORKSPACE, "gq_h_prepare: workspace %zu < %zu bytes", ws_bytes, need);
    const int64_t n = C, nblk = C / NB;
    float* A = reinterpret_cast<float*>(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    float* X = A + (size_t)n * n;
    float* Dinv = X + (size_t)n * n;
    uint8_t* dead = reinterpret_cast<uint8_t*>(Dinv + (size_t)n * NB);
    uint8_t* zc = dead + n;
    int rc;

    GQ_HIP(hipMemsetAsync(not_invertible, 0, sizeof(int), st));
    {
    ProfScope ps(PT_PREP_ELEM, st);
    hipLaunchKernelGGL(col_flags_kernel, dim3((unsigned)((C + 63) / 64)), dim3(256), 0, st, H, W, R, C, dead, zc);
    GQ_LAUNCH_CHECK();
    hipLaunchKernelGGL(zero_dead_cols_kernel, dim3(2048), dim3(256), 0, st, W, R, C, dead);
    GQ_LAUNCH_CHECK();
    hipLaunchKernelGGL(mask_h_kernel, dim3(4096), dim3(256), 0, st, H, C, zc);
    GQ_LAUNCH_CHECK();
    hipLaunchKernelGGL(damp_kernel, dim3(1), dim3(1024), 0, st, H, C, rel_damp);
    GQ_LAUNCH_CHECK();
    hipLaunchKernelGGL(reverse_copy_kernel, dim3(4096), dim3(256), 0, st, A, H, n);
    GQ_LAUNCH_CHECK();
    GQ_HIP(hipMemsetAsync(X, 0, (size_t)n * n * sizeof(float), st));
    }

    static bool attr_set = false;
    const size_t diag_lds = 2 * NB * NBP * sizeof(float);
    if (!attr_set) {
        GQ_HIP(hipFuncSetAttribute((const void*)diag_potrf_inv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)diag_lds));
        attr_set = true;
    }
    for (int64_t k = 0; k < nblk; ++k) {
        float* Akk = A + (k * NB) * n + k * NB;
        float* Dk = Dinv + k * NB * NB;
        {
            ProfScope ps(PT_DIAG_POTRF, st);
            hipLaunchKernelGGL(diag_potrf_inv_kernel, dim3(1), dim3(256), diag_lds, st, Akk, n, Dk, not_invertible);
            GQ_LAUNCH_CHECK();
        }
        const int64_t mrem = n - (k + 1) * NB;
        if (mrem > 0) {
            ProfScope ps(PT_CHOL_GEMM, st);
            float* A21 = A + ((k + 1) * NB) * n + k * NB;
            // A21 <- A21 * L11^-T   (in place: a workgroup owns whole rows, N == one tile)
            if ((rc = launch_gemm32<true, 1, false>(A21, n, A21, n, Dk, NB, mrem, NB, NB, st))) return rc;
            // A22 -= A21 A21^T, lower tiles only
            float* A22 = A + ((k + 1) * NB) * n + (k + 1) * NB;
            if ((rc = launch_gemm32<true, 0, true>(A22, n, A21, n, A21, n, mrem, mrem, NB, st))) return rc;
        }
    }
    {
        ProfScope ps(PT_TRTRI_GEMM, st);
        if ((rc = trtri_rec(A, X, U, Dinv, n, 0, nblk, st))) return rc;
    }
    ProfScope ps(PT_PREP_ELEM, st);
    hipLaunchKernelGGL(finish_u_kernel, dim3(4096), dim3(256), 0, st, U, X, n, not_invertible);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

}  // namespace gq
