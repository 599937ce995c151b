// gq_hessian.hip -- K1: H = beta*H + alpha * X^T X  (GPTQ.update, gptq.py:96,108-112).
//
// MFMA-bound SYRK.  fp16/bf16 activations: products of two 11-bit (8-bit)
// significands are exact in fp32, so v_mfma_f32_32x32x16_{f16,bf16} with fp32
// accumulation computes the same sum as the reference's fp32 addmm in a different
// order (tolerance-class, like the reference's own CPU-vs-CUDA difference).
// fp32 activations use v_mfma_f32_32x32x2_f32 (no TF32 on gfx950, and the reference
// switches TF32 off: gptq.py:24-25).
//
// Only the upper-triangular 128x128 tiles are computed; each is mirrored in the
// epilogue, so flops = T*C^2 (+ diagonal tiles) instead of 2*T*C^2.
//
// 16-bit path: X[T,C] is first transposed to Xt[C,Tp] (Tp = T rounded up to 32,
// zero padded) so that both MFMA operands are 16-byte K-contiguous fragments.
#include "gq_common.hpp"

namespace gq {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------- transpose
// Xt[c, t] = X[t, c] for 16-bit elements, 64x64 tiles through LDS.
__global__ __launch_bounds__(256) void transpose16_kernel(const uint16_t* __restrict__ X, int64_t T, int64_t C,
                                                          uint16_t* __restrict__ Xt, int64_t Tp) {
    __shared__ uint16_t tile[64][64 + 2];
    const int64_t t0 = (int64_t)blockIdx.x * 64, c0 = (int64_t)blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
    for (int r = ty; r < 64; r += 4) {
        int64_t t = t0 + r, c = c0 + tx;
        tile[r][tx] = (t < T && c < C) ? X[t * C + c] : (uint16_t)0;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        int64_t c = c0 + r, t = t0 + tx;
        if (c < C && t < Tp) Xt[c * Tp + t] = tile[tx][r];
    }
}

// ------------------------------------------------------------ 16-bit SYRK
constexpr int HT = 128;       // output tile
constexpr int HK = 32;        // k per stage (16-bit elements)
constexpr int HLD = HK + 8;   // LDS row stride in elements (80 B: keeps 16-B alignment)

template <bool BF16>
__device__ __forceinline__ f32x16 mfma16(const uint4& a, const uint4& b, f32x16 c) {
    if constexpr (BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                       0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0,
                                                      0);
}

__device__ __forceinline__ void tri_tile(int64_t bid, int64_t nt, int64_t& ti, int64_t& tj) {
    // linear index over the upper triangle, row-major: (0,0..nt-1),(1,1..nt-1),...
    int64_t i = 0, rem = bid;
    while (rem >= nt - i) {
        rem -= nt - i;
        ++i;
    }
    ti = i;
    tj = i + rem;
}

template <bool BF16>
__global__ __launch_bounds__(256) void syrk16_kernel(float* __restrict__ H, int64_t C, const uint16_t* __restrict__ Xt,
                                                     int64_t Tp, float beta, float alpha) {
    __shared__ __attribute__((aligned(16))) uint16_t As[HT * HLD];
    __shared__ __attribute__((aligned(16))) uint16_t Bs[HT * HLD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int64_t nt = (C + HT - 1) / HT;
    int64_t ti, tj;
    tri_tile(blockIdx.x, nt, ti, tj);
    const int64_t i0 = ti * HT, j0 = tj * HT;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int li = lane & 31, lk = lane >> 5;
    for (int64_t k0 = 0; k0 < Tp; k0 += HK) {
        // stage: 128 rows x 32 elements = 128 x 4 uint4 per operand; 2 per thread each
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            int idx = tid + t * 256;
            int rr = idx >> 2, c8 = (idx & 3) * 8;
            uint4 va = make_uint4(0, 0, 0, 0), vb = make_uint4(0, 0, 0, 0);
            if (i0 + rr < C) va = *reinterpret_cast<const uint4*>(Xt + (i0 + rr) * Tp + k0 + c8);
            if (j0 + rr < C) vb = *reinterpret_cast<const uint4*>(Xt + (j0 + rr) * Tp + k0 + c8);
            *reinterpret_cast<uint4*>(As + rr * HLD + c8) = va;
            *reinterpret_cast<uint4*>(Bs + rr * HLD + c8) = vb;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < HK; kk += 16) {
            uint4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                a[i] = *reinterpret_cast<const uint4*>(As + (wm * 64 + i * 32 + li) * HLD + kk + lk * 8);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                b[j] = *reinterpret_cast<const uint4*>(Bs + (wn * 64 + j * 32 + li) * HLD + kk + lk * 8);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma16<BF16>(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
    const int lc = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t col = j0 + wn * 64 + j * 32 + lc;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t row = i0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (row < C && col < C) {
                    float h = beta * H[row * C + col] + alpha * acc[i][j][e];
                    H[row * C + col] = h;
                    if (ti != tj) H[col * C + row] = h;  // mirror (H stays exactly symmetric)
                }
            }
        }
}

// --------------------------------------------------------------- fp32 SYRK
// H tile = sum_t X[t, i] X[t, j]: both operands are read straight from X rows
// (lanes walk channels), no transpose needed for one-float MFMA operands.
constexpr int FK = 32;
constexpr int FLD = HT + 4;

__global__ __launch_bounds__(256) void syrk32_kernel(float* __restrict__ H, int64_t C, const float* __restrict__ X,
                                                     int64_t T, float beta, float alpha) {
    __shared__ __attribute__((aligned(16))) float As[FK * FLD];
    __shared__ __attribute__((aligned(16))) float Bs[FK * FLD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int64_t nt = (C + HT - 1) / HT;
    int64_t ti, tj;
    tri_tile(blockIdx.x, nt, ti, tj);
    const int64_t i0 = ti * HT, j0 = tj * HT;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    const int li = lane & 31, lk = lane >> 5;
    for (int64_t k0 = 0; k0 < T; k0 += FK) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            int idx = tid + t * 256;  // 32 rows x 32 float4
            int kk = idx >> 5, c4 = (idx & 31) * 4;
            float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
            if (k0 + kk < T) {
                const float* pa = X + (k0 + kk) * C + i0 + c4;
                const float* pb = X + (k0 + kk) * C + j0 + c4;
                if (i0 + c4 + 3 < C) va = *reinterpret_cast<const float4*>(pa);
                if (j0 + c4 + 3 < C) vb = *reinterpret_cast<const float4*>(pb);
            }
            *reinterpret_cast<float4*>(As + kk * FLD + c4) = va;
            *reinterpret_cast<float4*>(Bs + kk * FLD + c4) = vb;
        }
        __syncthreads();
#pragma unroll 4
        for (int kk = 0; kk < FK; kk += 2) {
            float av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = As[(kk + lk) * FLD + wm * 64 + i * 32 + li];
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = Bs[(kk + lk) * FLD + wn * 64 + j * 32 + li];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    const int lc = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t col = j0 + wn * 64 + j * 32 + lc;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t row = i0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (row < C && col < C) {
                    float h = beta * H[row * C + col] + alpha * acc[i][j][e];
                    H[row * C + col] = h;
                    if (ti != tj) H[col * C + row] = h;
                }
            }
        }
}

size_t h_accumulate_workspace_bytes(int64_t T, int64_t C) {
    const int64_t Tp = (T + HK - 1) / HK * HK;
    return (size_t)C * (size_t)Tp * 2 + 256;
}

int h_accumulate(float* H, const void* X, int x_dtype, int64_t T, int64_t C, float beta, float alpha, void* ws,
                 size_t ws_bytes, hipStream_t st) {
    if (!H || !X) GQ_FAIL(GQ_E_NULL, "gq_h_accumulate: null pointer");
    if (T <= 0 || C <= 0 || (C % 8)) GQ_FAIL(GQ_E_BAD_SHAPE, "gq_h_accumulate: T=%ld C=%ld (C %% 8 != 0)", (long)T, (long)C);
    const int64_t nt = (C + HT - 1) / HT;
    const dim3 grid((unsigned)(nt * (nt + 1) / 2)), block(256);
    if (x_dtype == GQ_F32) {
        ProfScope ps(PT_SYRK, st);
        hipLaunchKernelGGL(syrk32_kernel, grid, block, 0, st, H, C, (const float*)X, T, beta, alpha);
        GQ_LAUNCH_CHECK();
        return GQ_OK;
    }
    if (x_dtype != GQ_F16 && x_dtype != GQ_BF16) GQ_FAIL(GQ_E_BAD_TYPE, "gq_h_accumulate: unknown x_dtype %d", x_dtype);
    const size_t need = h_accumulate_workspace_bytes(T, C);
    if (!ws || ws_bytes < need) GQ_FAIL(GQ_E_WORKSPACE, "gq_h_accumulate: workspace %zu < %zu bytes", ws_bytes, need);
    const int64_t Tp = (T + HK - 1) / HK * HK;
    uint16_t* Xt = reinterpret_cast<uint16_t*>(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    dim3 tg((unsigned)((Tp + 63) / 64), (unsigned)((C + 63) / 64));
    {
        ProfScope ps(PT_TRANSPOSE, st);
        hipLaunchKernelGGL(transpose16_kernel, tg, block, 0, st, (const uint16_t*)X, T, C, Xt, Tp);
        GQ_LAUNCH_CHECK();
    }
    ProfScope ps(PT_SYRK, st);
    if (x_dtype == GQ_BF16)
        hipLaunchKernelGGL(syrk16_kernel<true>, grid, block, 0, st, H, C, Xt, Tp, beta, alpha);
    else
        hipLaunchKernelGGL(syrk16_kernel<false>, grid, block, 0, st, H, C, Xt, Tp, beta, alpha);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

}  // namespace gq
