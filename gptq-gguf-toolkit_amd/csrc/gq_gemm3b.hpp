// gq_gemm3b.hpp -- fp32-accurate GEMM on the bf16 matrix cores ("bf16x3" split), for the large
// GEMMs of the blocked Cholesky / triangular inverse (K3) only.
//
// v_mfma_f32_32x32x2_f32 peaks at 157 TFLOP/s, v_mfma_f32_32x32x16_bf16 at 2.5 PFLOP/s.  Every fp32
// operand is split EXACTLY into three bf16 terms by TRUNCATION, x = x1 + x2 + x3 (x1 = the top 8
// significand bits of x, x2 = the top 8 of x - x1, x3 = x - x1 - x2: 24 = 8 + 8 + 8, every subtraction is
// exact and x3 fits in bf16), and a*b is accumulated in fp32 from the six products whose weight is >= 2^-16:
// a1b3, a3b1, a2b2, a1b2, a2b1, a1b1 (smallest first; each bf16 x bf16 product is exact in fp32).  The
// dropped terms are <= 2^-22 |a b|, the size of an fp32 rounding, so the result has fp32-GEMM accuracy at
// 6/16 of the MFMA cost of the fp32 instruction.  NOT bit-identical to gq_gemm32.hpp: the GPTQ trailing update (parity gate: bit-exact
// against the reference's sgemm chain) never uses this file; U = chol(H^-1) is checked in fp64
// with an fp32 tolerance (tests/test_gpu_parity.py).
//
// Same interface and modes as gemm32_kernel (MODE 0/1/2, TRANS_B, LOWER, KR; no CHAIN).
// Workgroup = 256 threads = 4 waves (2x2), tile 128x128, wave tile 64x64 = 2x2 MFMA tiles.  K streams in
// stages of 32: fp32 global loads into registers one stage ahead, split on the VALU, stored as three
// bf16 planes per operand ([128 rows][32 k], 64-byte rows, 16-byte chunk kc of row r at kc ^ ((r>>2)&3):
// conflict-free ds_read_b128 fragments, as in the SYRK image) -- 48 KiB per workgroup, 3 workgroups/CU.
#pragma once
#include "gq_common.hpp"
#include "gq_gemm32.hpp"

namespace gq {

typedef __bf16 g3_bf16x8 __attribute__((ext_vector_type(8)));
constexpr int G3_PLANE_BYTES = TM * TK * 2;           // 8 KiB
constexpr int G3_LDS_BYTES = 6 * G3_PLANE_BYTES;      // A1 A2 A3 B1 B2 B3

// two fp32 -> three dwords, each holding the bf16 pair (x, y) of one plane (x in the low half).
// Truncation split: 2 ANDs + 2 SUBs per element, 3 v_perm_b32 per pair.
__device__ __forceinline__ void g3_split2(float x, float y, unsigned& p1, unsigned& p2, unsigned& p3) {
    const unsigned xb = __builtin_bit_cast(unsigned, x), yb = __builtin_bit_cast(unsigned, y);
    const unsigned x1 = xb & 0xffff0000u, y1 = yb & 0xffff0000u;
    const float xr = x - __builtin_bit_cast(float, x1), yr = y - __builtin_bit_cast(float, y1);
    const unsigned xrb = __builtin_bit_cast(unsigned, xr), yrb = __builtin_bit_cast(unsigned, yr);
    const unsigned x2 = xrb & 0xffff0000u, y2 = yrb & 0xffff0000u;
    const float xs = xr - __builtin_bit_cast(float, x2), ys = yr - __builtin_bit_cast(float, y2);  // <= 8 significant bits
    // v_perm_b32(hi_src, lo_src, sel): bytes 0-3 index lo_src, 4-7 hi_src; take the upper halves
    p1 = __builtin_amdgcn_perm(y1, x1, 0x07060302u);
    p2 = __builtin_amdgcn_perm(y2, x2, 0x07060302u);
    p3 = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, ys), __builtin_bit_cast(unsigned, xs), 0x07060302u);
}

// [rows][32 k] fp32 chunk held as 4 float4 per thread (g32_load_rows mapping) -> three bf16 planes
__device__ __forceinline__ void g3_store_rows(const float4 (&v)[4], unsigned char* planes, int tid) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int idx = tid + t * 256, rr = idx >> 3, c4 = (idx & 7) * 4;
        unsigned a0, b0, c0, a1, b1, c1;
        g3_split2(v[t].x, v[t].y, a0, b0, c0);
        g3_split2(v[t].z, v[t].w, a1, b1, c1);
        const int off = rr * 64 + ((((c4 >> 3) ^ ((rr >> 2) & 3))) << 4) + ((c4 & 4) << 1);
        *reinterpret_cast<uint2*>(planes + off) = make_uint2(a0, a1);
        *reinterpret_cast<uint2*>(planes + G3_PLANE_BYTES + off) = make_uint2(b0, b1);
        *reinterpret_cast<uint2*>(planes + 2 * G3_PLANE_BYTES + off) = make_uint2(c0, c1);
    }
}
// g32_load_rows without edge predication: callers guarantee whole 128 x 32 chunks (M, N % 128 == 0, K % 32 == 0)
__device__ __forceinline__ void g3_load_rows_full(float4 (&v)[4], const float* P, int64_t ld, int64_t r0, int64_t k0, int tid) {
    const float* p = P + (r0 + (tid >> 3)) * ld + k0 + (tid & 7) * 4;
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = *reinterpret_cast<const float4*>(p + (int64_t)t * 32 * ld);
}
__device__ __forceinline__ void g3_load_kn_full(float (&v)[16], const float* B, int64_t ldb, int64_t n0, int64_t k0, int tid) {
    const float* p = B + (k0 + (tid >> 7) * 16) * ldb + n0 + (tid & 127);
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = p[(int64_t)e * ldb];
}
// [32 k][128 n] chunk of a [K,N] matrix: thread = (n = tid & 127, 16 consecutive k from (tid >> 7) * 16)
__device__ __forceinline__ void g3_load_kn(float (&v)[16], const float* B, int64_t ldb, int64_t n0, int64_t N, int64_t k0,
                                           int64_t K, int tid) {
    const int64_t n = n0 + (tid & 127);
    const int64_t kb = k0 + (tid >> 7) * 16;
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = (n < N && kb + e < K) ? B[(kb + e) * ldb + n] : 0.f;
}
__device__ __forceinline__ void g3_store_kn(const float (&v)[16], unsigned char* planes, int tid) {
    const int rr = tid & 127, kh = tid >> 7;
#pragma unroll
    for (int c = 0; c < 2; ++c) {  // two 16-byte chunks of 8 k
        unsigned a[4], b[4], d[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) g3_split2(v[c * 8 + 2 * e], v[c * 8 + 2 * e + 1], a[e], b[e], d[e]);
        const int off = rr * 64 + (((kh * 2 + c) ^ ((rr >> 2) & 3)) << 4);
        *reinterpret_cast<uint4*>(planes + off) = make_uint4(a[0], a[1], a[2], a[3]);
        *reinterpret_cast<uint4*>(planes + G3_PLANE_BYTES + off) = make_uint4(b[0], b[1], b[2], b[3]);
        *reinterpret_cast<uint4*>(planes + 2 * G3_PLANE_BYTES + off) = make_uint4(d[0], d[1], d[2], d[3]);
    }
}

template <bool TRANS_B, int MODE, bool LOWER, int KR = 0, bool FULL = false>
__global__ __launch_bounds__(256, TRANS_B ? 3 : 2) void gemm3b_kernel(float* Cmat, int64_t ldc, const float* A, int64_t lda,
                                                        const float* B, int64_t ldb, int64_t M, int64_t N, int64_t K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char g3_smem[];
    const unsigned bx = (KR == 1) ? gridDim.x - 1 - blockIdx.x : blockIdx.x;  // long tiles first
    const unsigned by = (KR == 3) ? gridDim.y - 1 - blockIdx.y : blockIdx.y;
    if (LOWER && bx > by) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int64_t m0 = (int64_t)by * TM, n0 = (int64_t)bx * TN;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    unsigned char* Ap = g3_smem;
    unsigned char* Bp = g3_smem + 3 * G3_PLANE_BYTES;
    float4 va[4];
    float4 vbt[TRANS_B ? 4 : 1];
    float vbn[TRANS_B ? 1 : 16];
    auto fetch = [&](int64_t k0) {
        if constexpr (FULL) {
            g3_load_rows_full(va, A, lda, m0, k0, tid);
            if constexpr (TRANS_B) g3_load_rows_full(vbt, B, ldb, n0, k0, tid);
            else g3_load_kn_full(vbn, B, ldb, n0, k0, tid);
        } else {
            g32_load_rows(va, A, lda, m0, M, k0, K, tid);
            if constexpr (TRANS_B) g32_load_rows(vbt, B, ldb, n0, N, k0, K, tid);
            else g3_load_kn(vbn, B, ldb, n0, N, k0, K, tid);
        }
    };
    auto commit = [&]() {
        g3_store_rows(va, Ap, tid);
        if constexpr (TRANS_B) g3_store_rows(vbt, Bp, tid);
        else g3_store_kn(vbn, Bp, tid);
    };
    const int li = lane & 31, lk = lane >> 5;
    int64_t kb = 0, ke = K;
    if constexpr (KR == 1) ke = (n0 + TN < K) ? n0 + TN : K;
    if constexpr (KR == 2) kb = (n0 < K) ? n0 : K;
    if constexpr (KR == 3) ke = (m0 + TM < K) ? m0 + TM : K;
    const int64_t nk = (ke - kb + TK - 1) / TK;
    // fragment byte offsets inside a plane (k16 step s adds the chunk pair 2s, 2s+1)
    int offA[2], offB[2], swA[2], swB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra = wm * 64 + i * 32 + li, rb = wn * 64 + i * 32 + li;
        offA[i] = ra * 64;
        swA[i] = (ra >> 2) & 3;
        offB[i] = rb * 64;
        swB[i] = (rb >> 2) & 3;
    }
    if (nk > 0) fetch(kb);
    for (int64_t t = 0; t < nk; ++t) {
        commit();
        __syncthreads();
        if (t + 1 < nk) fetch(kb + (t + 1) * TK);  // in flight during the MFMA block
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            g3_bf16x8 a[2][3], b[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    a[i][p] = *reinterpret_cast<const g3_bf16x8*>(Ap + p * G3_PLANE_BYTES + offA[i] + (((s2 * 2 + lk) ^ swA[i]) << 4));
                    b[i][p] = *reinterpret_cast<const g3_bf16x8*>(Bp + p * G3_PLANE_BYTES + offB[i] + (((s2 * 2 + lk) ^ swB[i]) << 4));
                }
            // six products per accumulator, smallest first; the four accumulators are interleaved so that
            // consecutive MFMAs never depend on each other
            constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
            for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[t6]], b[j][PB[t6]], acc[i][j], 0, 0, 0);
        }
        __syncthreads();  // every wave is done with this stage's planes
    }
    const int lc = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t col = n0 + wn * 64 + j * 32 + lc;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t rowi = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (FULL || (rowi < M && col < N)) {
                    float* p = Cmat + rowi * ldc + col;
                    if constexpr (MODE == 0) *p = *p - acc[i][j][e];
                    else if constexpr (MODE == 1) *p = acc[i][j][e];
                    else *p = -acc[i][j][e];
                }
            }
        }
}

template <bool TRANS_B, int MODE, bool LOWER, int KR = 0>
inline int launch_gemm3b(float* Cmat, int64_t ldc, const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M,
                         int64_t N, int64_t K, hipStream_t st) {
    if (M <= 0 || N <= 0 || K <= 0) return GQ_OK;
    if ((lda % 4) || (ldb % 4) || ((uintptr_t)A % 16) || ((uintptr_t)B % 16))
        GQ_FAIL(GQ_E_BAD_SHAPE, "gemm3b: A/B must be 16-byte aligned with ld %% 4 == 0");
    dim3 grid((unsigned)((N + TN - 1) / TN), (unsigned)((M + TM - 1) / TM)), block(256);
    if (M % TM == 0 && N % TN == 0 && K % TM == 0)  // whole tiles (k-ranges are 128-aligned too): no edge predication
        hipLaunchKernelGGL((gemm3b_kernel<TRANS_B, MODE, LOWER, KR, true>), grid, block, G3_LDS_BYTES, st, Cmat, ldc, A, lda,
                           B, ldb, M, N, K);
    else
        hipLaunchKernelGGL((gemm3b_kernel<TRANS_B, MODE, LOWER, KR, false>), grid, block, G3_LDS_BYTES, st, Cmat, ldc, A, lda,
                           B, ldb, M, N, K);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

}  // namespace gq
