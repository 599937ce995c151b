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
// stages of 16: fp32 global loads into registers two stages ahead, split on the VALU, stored as three
// bf16 planes per operand into a double-buffered LDS image (2 x 24 KiB per workgroup, up to 3 workgroups/CU).
#pragma once
#include "gq_common.hpp"
#include "gq_gemm32.hpp"

namespace gq {

typedef __bf16 g3_bf16x8 __attribute__((ext_vector_type(8)));
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

// ---- stages of 16 k, double-buffered LDS image, double-buffered fragments ----
// One barrier per stage.  Stage t: 12 MFMAs | split + LDS write of chunk t+1 into the other image, global
// loads of chunk t+2 | barrier | fragment reads of chunk t+1 | 12 MFMAs -- the VALU split, the LDS traffic and
// the loads all issue in the shadow of this wave's MFMAs.  Planes are [128 rows][16 k] bf16 (32-byte rows: a
// wave's 64 fragment reads cover 1 KiB contiguously, no swizzle needed).  Same six products in the same order
// per accumulator and k16 step whatever the staging.
constexpr int G3_TK = 16;
constexpr int G3_PLANE_BYTES = TM * G3_TK * 2;        // 4 KiB
constexpr int G3_STAGE_BYTES = 6 * G3_PLANE_BYTES;    // 24 KiB
constexpr int G3_LDS_BYTES = 2 * G3_STAGE_BYTES;      // 48 KiB: two workgroups per CU

// [128 rows][16 k] fp32 chunk: 2 float4 per thread (row = idx >> 2, k = (idx & 3) * 4)
template <bool FULL>
__device__ __forceinline__ void g3_load_rows(float4 (&v)[2], const float* P, int64_t ld, int64_t r0, int64_t rmax,
                                              int64_t k0, int64_t K, int tid) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int idx = tid + t * 256, rr = idx >> 2, c4 = (idx & 3) * 4;
        const float* p = P + (r0 + rr) * ld + k0 + c4;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (FULL) {
            x = *reinterpret_cast<const float4*>(p);
        } else if (r0 + rr < rmax) {
            if (k0 + c4 + 3 < K) x = *reinterpret_cast<const float4*>(p);
            else {
                if (k0 + c4 + 0 < K) x.x = p[0];
                if (k0 + c4 + 1 < K) x.y = p[1];
                if (k0 + c4 + 2 < K) x.z = p[2];
            }
        }
        v[t].x = x.x; v[t].y = x.y; v[t].z = x.z; v[t].w = x.w;
    }
}
__device__ __forceinline__ void g3_store_rows(const float4 (&v)[2], unsigned char* planes, int tid) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int idx = tid + t * 256, rr = idx >> 2, c4 = (idx & 3) * 4;
        unsigned a0, b0, c0, a1, b1, c1;
        g3_split2(v[t].x, v[t].y, a0, b0, c0);
        g3_split2(v[t].z, v[t].w, a1, b1, c1);
        const int off = (c4 >> 3) * (TM * 16) + rr * 16 + (c4 & 4) * 2;
        *reinterpret_cast<uint2*>(planes + off) = make_uint2(a0, a1);
        *reinterpret_cast<uint2*>(planes + G3_PLANE_BYTES + off) = make_uint2(b0, b1);
        *reinterpret_cast<uint2*>(planes + 2 * G3_PLANE_BYTES + off) = make_uint2(c0, c1);
    }
}
// [16 k][128 n] chunk of a [K,N] matrix: thread = (n = tid & 127, 8 consecutive k from (tid >> 7) * 8)
template <bool FULL>
__device__ __forceinline__ void g3_load_kn(float (&v)[8], const float* B, int64_t ldb, int64_t n0, int64_t N, int64_t k0,
                                            int64_t K, int tid) {
    const int64_t n = n0 + (tid & 127);
    const int64_t kb = k0 + (tid >> 7) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (FULL || (n < N && kb + e < K)) ? B[(kb + e) * ldb + n] : 0.f;
}
__device__ __forceinline__ void g3_store_kn(const float (&v)[8], unsigned char* planes, int tid) {
    const int rr = tid & 127, kh = tid >> 7;
    unsigned a[4], b[4], d[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) g3_split2(v[2 * e], v[2 * e + 1], a[e], b[e], d[e]);
    const int off = kh * (TM * 16) + rr * 16;
    *reinterpret_cast<uint4*>(planes + off) = make_uint4(a[0], a[1], a[2], a[3]);
    *reinterpret_cast<uint4*>(planes + G3_PLANE_BYTES + off) = make_uint4(b[0], b[1], b[2], b[3]);
    *reinterpret_cast<uint4*>(planes + 2 * G3_PLANE_BYTES + off) = make_uint4(d[0], d[1], d[2], d[3]);
}

template <bool TRANS_B, int MODE, bool LOWER, int KR = 0, bool FULL = false>
__global__ __launch_bounds__(256, 2) void gemm3b_kernel(float* Cmat, int64_t ldc, const float* A, int64_t lda,
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

    float4 va[2];
    float4 vbt[TRANS_B ? 2 : 1];
    float vbn[TRANS_B ? 1 : 8];
    auto fetch = [&](int64_t k0) {
        g3_load_rows<FULL>(va, A, lda, m0, M, k0, K, tid);
        if constexpr (TRANS_B) g3_load_rows<FULL>(vbt, B, ldb, n0, N, k0, K, tid);
        else g3_load_kn<FULL>(vbn, B, ldb, n0, N, k0, K, tid);
    };
    auto commit = [&](int buf) {
        unsigned char* Ap = g3_smem + buf * G3_STAGE_BYTES;
        unsigned char* Bp = Ap + 3 * G3_PLANE_BYTES;
        g3_store_rows(va, Ap, tid);
        if constexpr (TRANS_B) g3_store_rows(vbt, Bp, tid);
        else g3_store_kn(vbn, Bp, tid);
    };
    const int li = lane & 31, lk = lane >> 5;
    int64_t kb = 0, ke = K;
    if constexpr (KR == 1) ke = (n0 + TN < K) ? n0 + TN : K;
    if constexpr (KR == 2) kb = (n0 < K) ? n0 : K;
    if constexpr (KR == 3) ke = (m0 + TM < K) ? m0 + TM : K;
    const int64_t nk = (ke - kb + G3_TK - 1) / G3_TK;
    int offA[2], offB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        offA[i] = lk * (TM * 16) + (wm * 64 + i * 32 + li) * 16;
        offB[i] = 3 * G3_PLANE_BYTES + lk * (TM * 16) + (wn * 64 + i * 32 + li) * 16;
    }
    g3_bf16x8 fa[2][2][3], fb[2][2][3];  // [set][tile][plane]
    auto load_frags = [&](int buf, g3_bf16x8 (&a)[2][3], g3_bf16x8 (&b)[2][3]) {
        const unsigned char* S = g3_smem + buf * G3_STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[i][p] = *reinterpret_cast<const g3_bf16x8*>(S + p * G3_PLANE_BYTES + offA[i]);
                b[i][p] = *reinterpret_cast<const g3_bf16x8*>(S + p * G3_PLANE_BYTES + offB[i]);
            }
    };
    // six products per accumulator, smallest first; the four accumulators are interleaved so that
    // consecutive MFMAs never depend on each other
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
    auto mfmas = [&](int t_lo, int t_hi, const g3_bf16x8 (&a)[2][3], const g3_bf16x8 (&b)[2][3]) {
#pragma unroll
        for (int t6 = t_lo; t6 < t_hi; ++t6)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[t6]], b[j][PB[t6]], acc[i][j], 0, 0, 0);
    };
    auto clampk = [&](int64_t t) { return kb + ((t < nk) ? t : nk - 1) * G3_TK; };
    if (nk > 0) {
        fetch(kb);
        commit(0);
        fetch(clampk(1));
        __syncthreads();
        load_frags(0, fa[0], fb[0]);
    }
    // stage body for a compile-time parity (fragment sets and LDS images alternate)
#define GQ_G3_STAGE(PAR, T)                                                          \
    do {                                                                              \
        mfmas(0, 3, fa[PAR], fb[PAR]);                                                \
        commit((PAR) ^ 1); /* past the end: the last chunk again, into the idle image */ \
        fetch(clampk((T) + 2));                                                       \
        __syncthreads();                                                              \
        load_frags((PAR) ^ 1, fa[(PAR) ^ 1], fb[(PAR) ^ 1]);                          \
        mfmas(3, 6, fa[PAR], fb[PAR]);                                                \
    } while (0)
    for (int64_t t = 0; t < nk; t += 2) {
        GQ_G3_STAGE(0, t);
        if (t + 1 < nk) GQ_G3_STAGE(1, t + 1);
    }
#undef GQ_G3_STAGE
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
    const bool full = (M % TM == 0 && N % TN == 0 && K % TM == 0);  // whole tiles (k-ranges are 128-aligned too)
    if (full)
        hipLaunchKernelGGL((gemm3b_kernel<TRANS_B, MODE, LOWER, KR, true>), grid, block, G3_LDS_BYTES, st, Cmat, ldc, A,
                           lda, B, ldb, M, N, K);
    else
        hipLaunchKernelGGL((gemm3b_kernel<TRANS_B, MODE, LOWER, KR, false>), grid, block, G3_LDS_BYTES, st, Cmat, ldc, A,
                           lda, B, ldb, M, N, K);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

}  // namespace gq
