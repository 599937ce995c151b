// gq_gemm32.hpp -- fp32 GEMM on v_mfma_f32_32x32x2_f32 (exact fp32: every output is a
// k-ordered fmaf chain), shared by the GPTQ trailing update (K6) and the blocked
// Cholesky / triangular inverse (K3).
//
//   MODE 0:  C = C - A op(B)      (acc chain starts at 0, then ONE subtraction)
//   MODE 1:  C = A op(B)
//   MODE 2:  C = -(A op(B))
//   TRANS_B: op(B) = B^T with B stored [N,K] row-major (else B is [K,N] row-major)
//   LOWER:   only tiles with tile_row >= tile_col are computed (square, symmetric updates)
//
// Workgroup = 256 threads = 4 waves (2x2); tile 128x128; each wave owns 64x64 = 2x2
// MFMA tiles (64 accumulator VGPRs); K streams through LDS in chunks of 32.
#pragma once
#include "gq_common.hpp"

namespace gq {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int TM = 128, TN = 128, TK = 32;
constexpr int LDA_S = TK + 1;   // A tile [TM][TK]: lanes walk rows -> odd stride
constexpr int LDB_S = TN + 4;   // B tile [TK][TN] (NN): lanes walk columns
constexpr int LDBT_S = TK + 1;  // B tile [TN][TK] (NT)

template <bool TRANS_B, int MODE, bool LOWER>
__global__ __launch_bounds__(256) void gemm32_kernel(float* Cmat, int64_t ldc, const float* A, int64_t lda,
                                                     const float* B, int64_t ldb, int64_t M, int64_t N, int64_t K) {
    __shared__ float As[TM * LDA_S];
    __shared__ float Bs[TRANS_B ? TN * LDBT_S : TK * LDB_S];
    if (LOWER && blockIdx.x > blockIdx.y) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int64_t m0 = (int64_t)blockIdx.y * TM, n0 = (int64_t)blockIdx.x * TN;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    for (int64_t k0 = 0; k0 < K; k0 += TK) {
        // A[m0:m0+128, k0:k0+32]: 128 rows x 8 float4
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            int idx = tid + t * 256;
            int rr = idx >> 3, c4 = (idx & 7) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m0 + rr < M) {
                const float* p = A + (m0 + rr) * lda + k0 + c4;
                if (k0 + c4 + 3 < K) v = *reinterpret_cast<const float4*>(p);
                else {
                    if (k0 + c4 + 0 < K) v.x = p[0];
                    if (k0 + c4 + 1 < K) v.y = p[1];
                    if (k0 + c4 + 2 < K) v.z = p[2];
                }
            }
            float* o = As + rr * LDA_S + c4;
            o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
        }
        if constexpr (TRANS_B) {  // B[n0:n0+128, k0:k0+32]
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                int idx = tid + t * 256;
                int rr = idx >> 3, c4 = (idx & 7) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n0 + rr < N) {
                    const float* p = B + (n0 + rr) * ldb + k0 + c4;
                    if (k0 + c4 + 3 < K) v = *reinterpret_cast<const float4*>(p);
                    else {
                        if (k0 + c4 + 0 < K) v.x = p[0];
                        if (k0 + c4 + 1 < K) v.y = p[1];
                        if (k0 + c4 + 2 < K) v.z = p[2];
                    }
                }
                float* o = Bs + rr * LDBT_S + c4;
                o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
            }
        } else {  // B[k0:k0+32, n0:n0+128]: 32 rows x 32 float4
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                int idx = tid + t * 256;
                int kk = idx >> 5, c4 = (idx & 31) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k0 + kk < K) {
                    const float* p = B + (k0 + kk) * ldb + n0 + c4;
                    if (n0 + c4 + 3 < N) v = *reinterpret_cast<const float4*>(p);
                    else {
                        if (n0 + c4 + 0 < N) v.x = p[0];
                        if (n0 + c4 + 1 < N) v.y = p[1];
                        if (n0 + c4 + 2 < N) v.z = p[2];
                    }
                }
                *reinterpret_cast<float4*>(Bs + kk * LDB_S + c4) = v;
            }
        }
        __syncthreads();
        const int li = lane & 31, lk = lane >> 5;
#pragma unroll 4
        for (int kk = 0; kk < TK; kk += 2) {
            float av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = As[(wm * 64 + i * 32 + li) * LDA_S + kk + lk];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if constexpr (TRANS_B) bv[j] = Bs[(wn * 64 + j * 32 + li) * LDBT_S + kk + lk];
                else bv[j] = Bs[(kk + lk) * LDB_S + wn * 64 + j * 32 + li];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // D layout: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
    const int lc = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t col = n0 + wn * 64 + j * 32 + lc;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t rowi = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (rowi < M && col < N) {
                    float* p = Cmat + rowi * ldc + col;
                    if constexpr (MODE == 0) *p = *p - acc[i][j][e];
                    else if constexpr (MODE == 1) *p = acc[i][j][e];
                    else *p = -acc[i][j][e];
                }
            }
        }
}

template <bool TRANS_B, int MODE, bool LOWER>
inline int launch_gemm32(float* Cmat, int64_t ldc, const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M,
                         int64_t N, int64_t K, hipStream_t st) {
    if (M <= 0 || N <= 0 || K <= 0) return GQ_OK;
    if ((lda % 4) || (ldb % 4) || ((uintptr_t)A % 16) || ((uintptr_t)B % 16))
        GQ_FAIL(GQ_E_BAD_SHAPE, "gemm32: A/B must be 16-byte aligned with ld %% 4 == 0");
    dim3 grid((unsigned)((N + TN - 1) / TN), (unsigned)((M + TM - 1) / TM)), block(256);
    hipLaunchKernelGGL((gemm32_kernel<TRANS_B, MODE, LOWER>), grid, block, 0, st, Cmat, ldc, A, lda, B, ldb, M, N, K);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

}  // namespace gq
