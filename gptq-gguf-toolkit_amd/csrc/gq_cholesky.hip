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
#include <atomic>
#include <stdlib.h>

#include "gq_common.hpp"
#include "gq_gemm32.hpp"
#include "gq_gemm3b.hpp"
#include "gq_gemm3p.hpp"

#include <map>
#include <memory>
#include <mutex>

namespace gq {

constexpr int NB = 128;

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

// One workgroup per 64 columns: nothing but the 64 flags is touched when none of them is dead (the usual case).
__global__ __launch_bounds__(256) void zero_dead_cols_kernel(float* __restrict__ W, int64_t R, int64_t C,
                                                             const uint8_t* __restrict__ dead) {
    const int64_t j = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const bool dd = j < C && dead[j];
    if (!__syncthreads_or(dd)) return;
    if (dd)
        for (int64_t r = threadIdx.x >> 6; r < R; r += 4) W[r * C + j] = 0.0f;  // gptq.py:141
}

// Follower of a shared Hessian: apply the leader's dead-channel set to this W and compare this
// W's zero-column set with the leader's (equal sets => identical masked/damped H => identical U).
__global__ __launch_bounds__(256) void w_prepare_kernel(float* __restrict__ W, int64_t R, int64_t C,
                                                        const uint8_t* __restrict__ dead,
                                                        const uint8_t* __restrict__ zc, int* __restrict__ mismatch) {
    const int64_t j = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const int ty = threadIdx.x >> 6;
    __shared__ int nz[4][64];
    int any = 0;
    if (j < C) {
        const bool dd = dead[j];
        for (int64_t r = ty; r < R; r += 4) {
            if (dd) W[r * C + j] = 0.0f;  // gptq.py:141
            else any |= (W[r * C + j] != 0.0f);
        }
    }
    nz[ty][threadIdx.x & 63] = any;
    __syncthreads();
    if (ty == 0 && j < C) {
        int a = nz[0][threadIdx.x] | nz[1][threadIdx.x] | nz[2][threadIdx.x] | nz[3][threadIdx.x];
        uint8_t mine = dead[j] || !a;
        if (mine != zc[j]) atomicOr(mismatch, 1);
    }
}

// 16-byte variants of the two column scans (C % 4 == 0, 16-byte aligned W): a workgroup owns 64 columns as 16
// float4 lanes x 16 row lanes and keeps 8 rows per thread in flight -- 32 KB per workgroup instead of 1 KB, which
// is what a column scan with only C/64 workgroups needs to approach HBM speed (14336 x 4096: 712 -> ~150 us).
__global__ __launch_bounds__(256) void col_flags4_kernel(const float* __restrict__ H, const float* __restrict__ W,
                                                         int64_t R, int64_t C, uint8_t* __restrict__ dead,
                                                         uint8_t* __restrict__ zc) {
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t j = (int64_t)blockIdx.x * 64 + 4 * tx;
    __shared__ int nz[16][64];
    int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    if (j < C) {
#pragma unroll 8
        for (int64_t r = ty; r < R; r += 16) {
            const float4 v = *reinterpret_cast<const float4*>(W + r * C + j);
            a0 |= (v.x != 0.0f); a1 |= (v.y != 0.0f); a2 |= (v.z != 0.0f); a3 |= (v.w != 0.0f);
        }
    }
    nz[ty][4 * tx] = a0; nz[ty][4 * tx + 1] = a1; nz[ty][4 * tx + 2] = a2; nz[ty][4 * tx + 3] = a3;
    __syncthreads();
    const int64_t jc = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (threadIdx.x < 64 && jc < C) {
        int a = 0;
#pragma unroll
        for (int t = 0; t < 16; ++t) a |= nz[t][threadIdx.x];
        const uint8_t dd = H[jc * C + jc] == 0.0f;
        dead[jc] = dd;
        zc[jc] = dd || !a;
    }
}
__global__ __launch_bounds__(256) void w_prepare4_kernel(float* __restrict__ W, int64_t R, int64_t C,
                                                         const uint8_t* __restrict__ dead,
                                                         const uint8_t* __restrict__ zc, int* __restrict__ mismatch) {
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t j = (int64_t)blockIdx.x * 64 + 4 * tx;
    __shared__ int nz[16][64];
    int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    if (j < C) {
        const bool d0 = dead[j], d1 = dead[j + 1], d2 = dead[j + 2], d3 = dead[j + 3];
        const bool anyd = d0 | d1 | d2 | d3;
#pragma unroll 8
        for (int64_t r = ty; r < R; r += 16) {
            float4 v = *reinterpret_cast<const float4*>(W + r * C + j);
            if (anyd) {  // gptq.py:141
                if (d0) v.x = 0.0f;
                if (d1) v.y = 0.0f;
                if (d2) v.z = 0.0f;
                if (d3) v.w = 0.0f;
                *reinterpret_cast<float4*>(W + r * C + j) = v;
            }
            a0 |= (v.x != 0.0f); a1 |= (v.y != 0.0f); a2 |= (v.z != 0.0f); a3 |= (v.w != 0.0f);
        }
    }
    nz[ty][4 * tx] = a0; nz[ty][4 * tx + 1] = a1; nz[ty][4 * tx + 2] = a2; nz[ty][4 * tx + 3] = a3;
    __syncthreads();
    const int64_t jc = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (threadIdx.x < 64 && jc < C) {
        int a = 0;
#pragma unroll
        for (int t = 0; t < 16; ++t) a |= nz[t][threadIdx.x];
        const uint8_t mine = dead[jc] || !a;
        if (mine != zc[jc]) atomicOr(mismatch, 1);
    }
}

// H[zc,:] = 0; H[:,zc] = 0; H[zc,zc] = 1  (gptq.py:311-313; also covers H[dead,dead]=1, :135)
__global__ __launch_bounds__(256) void mask_h_kernel(float* __restrict__ H, int64_t C, const uint8_t* __restrict__ zc) {
    // workgroup b owns the indices 64 b .. 64 b + 63: for every flagged one it clears row and column and sets the
    // diagonal to 1.  Without flags (the usual case) a workgroup reads its 64 flags and returns.
    __shared__ int flagged[64];
    const int64_t i0 = (int64_t)blockIdx.x * 64;
    if (threadIdx.x < 64) flagged[threadIdx.x] = (i0 + threadIdx.x < C) && zc[i0 + threadIdx.x];
    __syncthreads();
    for (int t = 0; t < 64; ++t) {
        if (!flagged[t]) continue;  // workgroup-uniform
        const int64_t i = i0 + t;
        for (int64_t j = threadIdx.x; j < C; j += blockDim.x) {
            const float v = (i == j) ? 1.0f : 0.0f;
            H[i * C + j] = v;
            H[j * C + i] = v;
        }
    }
}

// H[dead, dead] = 1 alone (evopress/src/fast_obq.py:134-135: the damping there comes BEFORE the zero-column mask)
__global__ __launch_bounds__(256) void dead_diag_kernel(float* __restrict__ H, int64_t C, const uint8_t* __restrict__ dead) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < C && dead[j]) H[j * C + j] = 1.0f;
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

// Equilibration by powers of two: s_j = 2^-e_j with H_jj s_j^2 in [0.5, 2).  The chain factorises S H S (unit-order
// diagonal, entries <= 2) and U = U_hat S (finish_u_kernel).  Scaling by powers of two commutes with every fp32
// operation (no overflow / underflow at these magnitudes), so the fp32 and exact-split bf16 kernels compute bit for bit
// what they would without it; what it buys is that all k of a product weigh the same, which the row-scaled fp16 images
// of gq_gemm3p.hpp need for their error bound to mean something (a row of H spans sigma_max / sigma_min of the channels).
__global__ __launch_bounds__(256) void equil_kernel(const float* __restrict__ H, int64_t n, float* __restrict__ s) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const float h = H[j * n + j];
    float v = 1.0f;
    if (h > 0.0f && h < 3.0e38f) {
        int e;
        (void)frexpf(h, &e);                       // h = m 2^e, m in [0.5, 1)
        const int half = (e >= 0 ? e : e - 1) / 2;  // floor(e / 2): h 2^(-2 half) in [0.5, 2)
        v = ldexpf(1.0f, -half);
    }
    s[j] = v;
}
// A[i,j] = H[n-1-i, n-1-j] s[n-1-i] s[n-1-j] for the 128x128 blocks on and below the block diagonal (the chain never
// reads A above it).  One workgroup per row, 16-byte accesses on both sides (n % 128 == 0).
__global__ __launch_bounds__(256) void reverse_copy_kernel(float* __restrict__ A, const float* __restrict__ H,
                                                           int64_t n, const float* __restrict__ s) {
    const int64_t i = blockIdx.x, r = n - 1 - i;
    const int64_t jend = (i / 128 + 1) * 128;
    const float si = s ? s[r] : 1.0f;
    for (int64_t j = 4 * (int64_t)threadIdx.x; j < jend; j += 4 * 256) {
        const float4 h = *reinterpret_cast<const float4*>(H + r * n + (n - 4 - j));  // columns n-4-j .. n-1-j
        float4 a = make_float4(h.w, h.z, h.y, h.x);
        if (s) {
            const float4 sc = *reinterpret_cast<const float4*>(s + (n - 4 - j));
            a.x *= si * sc.w; a.y *= si * sc.z; a.z *= si * sc.y; a.w *= si * sc.x;
        }
        *reinterpret_cast<float4*>(A + i * n + j) = a;
    }
}

// U[i,j] = X[n-1-i, n-1-j] s[j] for j >= i, 0 below; identity if *flag.  One workgroup per row.
__global__ __launch_bounds__(256) void finish_u_kernel(float* __restrict__ U, const float* __restrict__ X, int64_t n,
                                                       const int* __restrict__ flag, const float* __restrict__ s) {
    const int64_t i = blockIdx.x, r = n - 1 - i;
    const bool bad = *flag != 0;
    for (int64_t j = 4 * (int64_t)threadIdx.x; j < n; j += 4 * 256) {
        float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bad) {  // gptq.py:321-323
            u.x = (j == i) ? 1.0f : 0.0f; u.y = (j + 1 == i) ? 1.0f : 0.0f;
            u.z = (j + 2 == i) ? 1.0f : 0.0f; u.w = (j + 3 == i) ? 1.0f : 0.0f;
        } else if (j + 3 >= i) {
            const float4 x = *reinterpret_cast<const float4*>(X + r * n + (n - 4 - j));
            float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
            if (s) sc = *reinterpret_cast<const float4*>(s + j);
            u.x = (j >= i) ? x.w * sc.x : 0.0f;
            u.y = (j + 1 >= i) ? x.z * sc.y : 0.0f;
            u.z = (j + 2 >= i) ? x.y * sc.z : 0.0f;
            u.w = x.x * sc.w;
        }
        *reinterpret_cast<float4*>(U + i * n + j) = u;
    }
}

// --------------------------------------------------- diagonal block: potrf + inverse
// One workgroup (256 threads), ONE barrier per column.  A_kk (lower) -> L_kk in place;
// Dinv = L_kk^-1 (dense 128x128, zeros above the diagonal).  Left-looking: in step j
// thread i (waves 0-1, one row each) forms a_ij - <L_i,0:j , L_j,0:j> with 16-byte LDS
// reads (its own row: conflict-free at a 132-float stride; the pivot row: broadcast) and
// recomputes the pivot's own dot product instead of waiting for it.  Waves 2-3 run the
// forward substitution for L^-1 one row behind the factorisation (thread c owns column c,
// stored transposed so that its dot products are 16-byte reads too).
// A non-positive (or NaN) pivot raises *flag (gptq.py:321-323 identity fallback).
constexpr int LDP = NB + 4;

__global__ __launch_bounds__(256) void diag_potrf_inv_kernel(float* __restrict__ A, int64_t lda,
                                                             float* __restrict__ Xout, int64_t ldx,
                                                             int* __restrict__ flag) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* S = smem;              // [NB][LDP]  L (lower), zeros above
    float* XT = smem + NB * LDP;  // [NB][LDP]  XT[c][i] = (L^-1)[i][c]
    float* dg = XT + NB * LDP;    // [NB] original diagonal
    const int tid = threadIdx.x;
    for (int idx = tid; idx < NB * NB; idx += 256) {
        int i = idx / NB, j = idx % NB;
        float v = (j <= i) ? A[i * lda + j] : 0.0f;
        S[i * LDP + j] = v;
        XT[i * LDP + j] = 0.0f;
        if (i == j) dg[i] = v;
    }
    __syncthreads();
    const bool fac = tid < NB;
    const int i = tid & (NB - 1);
    for (int j = 0; j <= NB; ++j) {
        if (fac) {
            if (j < NB && i >= j) {
                const float4* ri = reinterpret_cast<const float4*>(S + i * LDP);
                const float4* rj = reinterpret_cast<const float4*>(S + j * LDP);
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
                const int n4 = j >> 2;
                int p4 = 0;
                for (; p4 + 4 <= n4; p4 += 4) {  // 8 LDS reads in flight, 4 independent chains each
                    float4 a[4], b[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { a[u] = ri[p4 + u]; b[u] = rj[p4 + u]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        a0 = fmaf(a[u].x, b[u].x, a0); a1 = fmaf(a[u].y, b[u].y, a1);
                        a2 = fmaf(a[u].z, b[u].z, a2); a3 = fmaf(a[u].w, b[u].w, a3);
                        q0 = fmaf(b[u].x, b[u].x, q0); q1 = fmaf(b[u].y, b[u].y, q1);
                        q2 = fmaf(b[u].z, b[u].z, q2); q3 = fmaf(b[u].w, b[u].w, q3);
                    }
                }
                for (; p4 < n4; ++p4) {
                    const float4 a = ri[p4], b = rj[p4];
                    a0 = fmaf(a.x, b.x, a0); a1 = fmaf(a.y, b.y, a1); a2 = fmaf(a.z, b.z, a2); a3 = fmaf(a.w, b.w, a3);
                    q0 = fmaf(b.x, b.x, q0); q1 = fmaf(b.y, b.y, q1); q2 = fmaf(b.z, b.z, q2); q3 = fmaf(b.w, b.w, q3);
                }
                float acc = (a0 + a1) + (a2 + a3), accp = (q0 + q1) + (q2 + q3);
                for (int p = p4 * 4; p < j; ++p) {
                    const float a = S[i * LDP + p], b = S[j * LDP + p];
                    acc = fmaf(a, b, acc);
                    accp = fmaf(b, b, accp);
                }
                float piv = dg[j] - accp;
                const bool bad = !(piv > 0.0f);  // also NaN
                if (bad) piv = 1.0f;
                const float ljj = sqrtf(piv);
                if (i == j) {
                    S[j * LDP + j] = ljj;
                    if (bad) *flag = 1;
                } else {
                    S[i * LDP + j] = (S[i * LDP + j] - acc) / ljj;
                }
            }
        } else if (j >= 1) {
            const int r = j - 1, c = i;  // row r of L is final since the previous barrier
            if (c == r) {
                XT[c * LDP + r] = 1.0f / S[r * LDP + r];
            } else if (c < r) {
                const float4* lr = reinterpret_cast<const float4*>(S + r * LDP);
                const float4* xc = reinterpret_cast<const float4*>(XT + c * LDP);
                // XT[c][p] is 0 for p < c and for p >= r, S[r][p] is 0 for p > r: whole 16-byte chunks
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                int p4 = c >> 2;
                const int e4 = r >> 2;
                for (; p4 + 3 <= e4; p4 += 4) {
                    float4 a[4], b[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { a[u] = lr[p4 + u]; b[u] = xc[p4 + u]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        a0 = fmaf(a[u].x, b[u].x, a0); a1 = fmaf(a[u].y, b[u].y, a1);
                        a2 = fmaf(a[u].z, b[u].z, a2); a3 = fmaf(a[u].w, b[u].w, a3);
                    }
                }
                for (; p4 <= e4; ++p4) {
                    const float4 a = lr[p4], b = xc[p4];
                    a0 = fmaf(a.x, b.x, a0); a1 = fmaf(a.y, b.y, a1); a2 = fmaf(a.z, b.z, a2); a3 = fmaf(a.w, b.w, a3);
                }
                const float acc = (a0 + a1) + (a2 + a3);
                XT[c * LDP + r] = -acc / S[r * LDP + r];
            }
        }
        __syncthreads();
    }
    for (int idx = tid; idx < NB * NB; idx += 256) {
        int r = idx / NB, c = idx % NB;
        if (c <= r) A[r * lda + c] = S[r * LDP + c];
        Xout[r * ldx + c] = XT[c * LDP + r];
    }
}

// --------------------------------------- diagonal block, blocked variant (default)
// Same contract as diag_potrf_inv_kernel, ~5x shorter critical path: the 128x128 block is
// processed as 4x4 sub-blocks of 32x32.
//   * a 32x32 diagonal sub-block is factored AND inverted by ONE wave in registers: lane i holds
//     row i (32 VGPRs); the pivot row / column entries other lanes need are wave-uniform and come
//     through v_readlane (SGPR operands of the FMAs) -- no LDS, no barrier inside;
//   * the panel below it (P = A_panel * Dinv^T) and the trailing update of the remaining
//     sub-blocks (S_ij -= P_i P_j^T) are 32x32x32 products on v_mfma_f32_32x32x2_f32;
//   * L^-1 of the whole block is assembled from the sub-block inverses by block forward
//     substitution, again on the matrix cores (X_ij = -X_ii * sum_k L_ik X_kj).
// LDS rows have an odd stride (129 floats): MFMA operand reads walk rows with stride-1 banks.
constexpr int LDQ = NB + 1;
constexpr int TQ = 33;  // wave-private 32x32 scratch tile stride

__device__ __forceinline__ float rdlane(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// ---- 32x32 factor + inverse in registers (wave 0 of diag_blk_kernel): lane i holds row i of the block in a[0..31].
// Every update needs a broadcast L[c][j] = lane c's a[j]: v_readlane into an SGPR, then one VALU op.  Written as
// inline-asm batches of four (4 v_readlane, then 4 v_fma that each read an SGPR written >= 3 instructions earlier: no
// wait states): left to the compiler the 496 broadcasts of the factor loop are kept alive for the inverse loop, which
// needs the same values, and the SGPR file spills into VGPR lanes (v_writelane + s_nop per element: ~9 cycles per
// instruction measured, 23k cycles per 32x32 sub-block).
template <int C>
__device__ __forceinline__ void d_upd4(float (&a)[32], float aj) {  // a[C..C+3] -= aj * L[C..C+3][j]
    int t0, t1, t2, t3;
    asm volatile(
        "v_readlane_b32 %4, %8, %9\n\tv_readlane_b32 %5, %8, %10\n\tv_readlane_b32 %6, %8, %11\n\t"
        "v_readlane_b32 %7, %8, %12\n\t"
        "v_fma_f32 %0, -%8, %4, %0\n\tv_fma_f32 %1, -%8, %5, %1\n\tv_fma_f32 %2, -%8, %6, %2\n\tv_fma_f32 %3, -%8, %7, %3"
        : "+v"(a[C]), "+v"(a[C + 1]), "+v"(a[C + 2]), "+v"(a[C + 3]), "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3)
        : "v"(aj), "n"(C), "n"(C + 1), "n"(C + 2), "n"(C + 3));
}
template <int C>
__device__ __forceinline__ void d_upd8(float (&a)[32], float aj) {  // a[C..C+7] -= aj * L[C..C+7][j]
    int t0, t1, t2, t3, t4, t5, t6, t7;
    asm volatile(
        "v_readlane_b32 %8, %16, %17\n\tv_readlane_b32 %9, %16, %18\n\tv_readlane_b32 %10, %16, %19\n\t"
        "v_readlane_b32 %11, %16, %20\n\tv_readlane_b32 %12, %16, %21\n\tv_readlane_b32 %13, %16, %22\n\t"
        "v_readlane_b32 %14, %16, %23\n\tv_readlane_b32 %15, %16, %24\n\t"
        "v_fma_f32 %0, -%16, %8, %0\n\tv_fma_f32 %1, -%16, %9, %1\n\tv_fma_f32 %2, -%16, %10, %2\n\t"
        "v_fma_f32 %3, -%16, %11, %3\n\tv_fma_f32 %4, -%16, %12, %4\n\tv_fma_f32 %5, -%16, %13, %5\n\t"
        "v_fma_f32 %6, -%16, %14, %6\n\tv_fma_f32 %7, -%16, %15, %7"
        : "+v"(a[C]), "+v"(a[C + 1]), "+v"(a[C + 2]), "+v"(a[C + 3]), "+v"(a[C + 4]), "+v"(a[C + 5]), "+v"(a[C + 6]),
          "+v"(a[C + 7]), "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7)
        : "v"(aj), "n"(C), "n"(C + 1), "n"(C + 2), "n"(C + 3), "n"(C + 4), "n"(C + 5), "n"(C + 6), "n"(C + 7));
}
template <int C>
__device__ __forceinline__ void d_upd1(float (&a)[32], float aj) {
    int t0;
    asm volatile("v_readlane_b32 %1, %2, %3\n\ts_nop 1\n\tv_fma_f32 %0, -%2, %1, %0" : "+v"(a[C]), "=&s"(t0) : "v"(aj), "n"(C));
}
template <int C>
__device__ __forceinline__ void d_updates(float (&a)[32], float aj) {  // columns C..31
    if constexpr (C + 7 <= 31) {
        d_upd8<C>(a, aj);
        d_updates<C + 8>(a, aj);
    } else if constexpr (C + 3 <= 31) {
        d_upd4<C>(a, aj);
        d_updates<C + 4>(a, aj);
    } else if constexpr (C <= 31) {
        d_upd1<C>(a, aj);
        d_updates<C + 1>(a, aj);
    }
}
template <int J>
__device__ __forceinline__ void d_factor(float (&a)[32], int i, float& myinv, bool& bad) {
    asm volatile("s_nop 1" : "+v"(a[J]));  // a[J] may be the last VGPR written inside the previous step's asm
    float pj = rdlane(a[J], J);
    if (!(pj > 0.0f)) {  // wave-uniform; also NaN
        bad = true;
        pj = 1.0f;
    }
    // 1/sqrt by v_rsq_f32 + one Newton step (<= 2 ulp) instead of an IEEE sqrt and an IEEE division: the 128
    // pivots of a block are a serial chain, ~35 dependent instructions shorter each this way.  U is a
    // tolerance-class output: the pivot itself carries the rounding of a 14336-term sum.
    float inv = __builtin_amdgcn_rsqf(pj);
    inv = fmaf(inv, fmaf(-0.5f * pj * inv, inv, 0.5f), inv);
    const float ljj = pj * inv;
    // (lane == J) masks are loop-invariant: the compiler would precompute all 64 of them, run out of SGPRs and
    // spill them into VGPR lanes; an opaque copy of the lane index makes it compare in place (one v_cmp)
    int ii = i;
    asm volatile("" : "+v"(ii));
    const bool mine = ii == J;
    if (mine) myinv = inv;
    a[J] = mine ? ljj : a[J] * inv;
    // the asm below starts with a v_readlane of the VGPR the VALU has just written: the compiler's hazard
    // recogniser cannot see into the asm text (measured: wrong pivots without these wait states)
    asm volatile("s_nop 1" : "+v"(a[J]));
    d_updates<J + 1>(a, a[J]);
    if constexpr (J < 31) d_factor<J + 1>(a, i, myinv, bad);
}
// acc[k] += L[R][P+k] * x[P+k]: four independent chains (the dot product of row R of L with the inverse column)
template <int R, int P>
__device__ __forceinline__ void d_dot4(const float (&a)[32], const float (&x)[32], float (&acc)[4]) {
    int t0, t1, t2, t3;
    asm volatile(
        "v_readlane_b32 %4, %8, %16\n\tv_readlane_b32 %5, %9, %16\n\tv_readlane_b32 %6, %10, %16\n\t"
        "v_readlane_b32 %7, %11, %16\n\t"
        "v_fma_f32 %0, %4, %12, %0\n\tv_fma_f32 %1, %5, %13, %1\n\tv_fma_f32 %2, %6, %14, %2\n\tv_fma_f32 %3, %7, %15, %3"
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3)
        : "v"(a[P]), "v"(a[P + 1]), "v"(a[P + 2]), "v"(a[P + 3]), "v"(x[P]), "v"(x[P + 1]), "v"(x[P + 2]), "v"(x[P + 3]),
          "n"(R));
}
template <int R, int P>
__device__ __forceinline__ void d_dot8(const float (&a)[32], const float (&x)[32], float (&acc)[4]) {
    int t0, t1, t2, t3, t4, t5, t6, t7;
    asm volatile(
        "v_readlane_b32 %4, %12, %28\n\tv_readlane_b32 %5, %13, %28\n\tv_readlane_b32 %6, %14, %28\n\t"
        "v_readlane_b32 %7, %15, %28\n\tv_readlane_b32 %8, %16, %28\n\tv_readlane_b32 %9, %17, %28\n\t"
        "v_readlane_b32 %10, %18, %28\n\tv_readlane_b32 %11, %19, %28\n\t"
        "v_fma_f32 %0, %4, %20, %0\n\tv_fma_f32 %1, %5, %21, %1\n\tv_fma_f32 %2, %6, %22, %2\n\t"
        "v_fma_f32 %3, %7, %23, %3\n\tv_fma_f32 %0, %8, %24, %0\n\tv_fma_f32 %1, %9, %25, %1\n\t"
        "v_fma_f32 %2, %10, %26, %2\n\tv_fma_f32 %3, %11, %27, %3"
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4),
          "=&s"(t5), "=&s"(t6), "=&s"(t7)
        : "v"(a[P]), "v"(a[P + 1]), "v"(a[P + 2]), "v"(a[P + 3]), "v"(a[P + 4]), "v"(a[P + 5]), "v"(a[P + 6]), "v"(a[P + 7]),
          "v"(x[P]), "v"(x[P + 1]), "v"(x[P + 2]), "v"(x[P + 3]), "v"(x[P + 4]), "v"(x[P + 5]), "v"(x[P + 6]), "v"(x[P + 7]),
          "n"(R));
}
template <int R, int P>
__device__ __forceinline__ void d_dot1(const float (&a)[32], const float (&x)[32], float& acc) {
    int t0;
    asm volatile("v_readlane_b32 %1, %2, %4\n\ts_nop 1\n\tv_fma_f32 %0, %1, %3, %0" : "+v"(acc), "=&s"(t0) : "v"(a[P]), "v"(x[P]), "n"(R));
}
template <int R, int P>
__device__ __forceinline__ void d_dot(const float (&a)[32], const float (&x)[32], float (&acc)[4]) {  // p = P..R-1
    if constexpr (P + 7 < R) {
        d_dot8<R, P>(a, x, acc);
        d_dot<R, P + 8>(a, x, acc);
    } else if constexpr (P + 3 < R) {
        d_dot4<R, P>(a, x, acc);
        d_dot<R, P + 4>(a, x, acc);
    } else if constexpr (P < R) {
        d_dot1<R, P>(a, x, acc[P & 3]);
        d_dot<R, P + 1>(a, x, acc);
    }
}
template <int R>
__device__ __forceinline__ void d_inverse(const float (&a)[32], float (&x)[32], int i, float myinv) {  // lane i = column i
    const float irr = rdlane(myinv, R);
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    d_dot<R, 0>(a, x, acc);
    const float dot = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    int ii = i;
    asm volatile("" : "+v"(ii));
    x[R] = (R < ii) ? 0.0f : ((R == ii) ? irr : -dot * irr);
    if constexpr (R < 31) d_inverse<R + 1>(a, x, i, myinv);
}

// C[32x32] (accumulator) = sum_k Arow[i][k] * Brow[j][k], k < 32*nk32: both operands row-major in LDS
// (lane (i = l&31, kh = l>>5) reads A[i][2p+kh] and B[j=l&31][2p+kh]); strides in floats.
__device__ __forceinline__ f32x16 mfma_nt_32(const float* Ap, int lda_, const float* Bp, int ldb_, int lane) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    const int li = lane & 31, lk = lane >> 5;
    // all 32 operand reads are issued before the first MFMA: one LDS latency per product instead of 16
    float av[16], bv[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        av[p] = Ap[li * lda_ + 2 * p + lk];
        bv[p] = Bp[li * ldb_ + 2 * p + lk];
    }
#pragma unroll
    for (int p = 0; p < 16; ++p) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[p], bv[p], acc, 0, 0, 0);
    return acc;
}

// (diag_blk_kernel, the four-wave leaf of r01 / r02 that used the helpers above with ONE wave factoring and inverting each
// 32x32 sub-block -- 51.5 us per leaf against 34.3 us for the five-wave pipeline of gq_diag5.hpp -- lives in the git history.)

}  // namespace gq
#include "gq_diag5.hpp"
namespace gq {

// ---- the top recursion levels on pre-split operand images (gq_gemm3p.hpp) ----
// A node (n1 | n2) goes there when both halves are multiples of 256 and at least p3_min() wide.  The schedules of all
// its GEMMs depend on C only: they are planned once per C (cached), and uploaded with ONE copy per gq_h_prepare
// call (the upload also zeroes the pop counters).
// (The switches are read on every call -- a handful per transformer block -- so that tests can flip them.)
static int64_t p3_min() { return opt(OPT_chol_3p_min); }  // 0: never
static int p3_planes() { return opt(OPT_chol_planes) == 3 ? 3 : 2; }  // default: row-scaled fp16 x 2, equilibrated matrix
static inline bool p3_node(int64_t n1, int64_t n2) {
    return p3_min() > 0 && n1 >= p3_min() && n2 >= p3_min() && n1 % 256 == 0 && n2 % 256 == 0;
}
constexpr int P3_MAX_SLOTS = 1024;  // 256 MiB of K-split partial sums
struct P3Gemm {
    size_t table_at, rlist_at;  // word offsets into the uploaded buffer
    int n_reduce;
};
struct P3Plans {
    std::vector<uint32_t> words;
    std::vector<P3Gemm> gemms;  // three launches per node, in the order chol_inv_rec issues them
};
static void p3_collect(P3Plans& pl, int64_t lo, int64_t hi) {
    if (hi - lo == 1) return;
    const int64_t mid = (lo + hi) / 2;
    const int64_t n1 = (mid - lo) * NB, n2 = (hi - mid) * NB;
    p3_collect(pl, lo, mid);
    const bool big = p3_node(n1, n2);
    const int gran = p3_planes() == 3 ? 2 : 4;
    auto add = [&](const std::vector<p3::GemmShape>& shs) {
        p3::Plan p = p3::make_plan(shs, gran, P3_MAX_SLOTS);
        P3Gemm g;
        while (pl.words.size() % 4) pl.words.push_back(0u);
        g.table_at = pl.words.size();
        pl.words.insert(pl.words.end(), p.table.begin(), p.table.end());
        g.rlist_at = pl.words.size();
        pl.words.insert(pl.words.end(), p.rlist.begin(), p.rlist.end());
        g.n_reduce = (int)(p.rlist.size() / 2);
        if (p.nslots < 0) fprintf(stderr, "gq: image GEMM plan needs more than %d partial slots\n", P3_MAX_SLOTS), abort();
        pl.gemms.push_back(g);
    };
    if (big) {
        const int t1 = (int)(n1 / 256), t2 = (int)(n2 / 256), k1 = (int)(n1 / 32);
        add({{t2, t1, k1, 1, false}});                          // L21 = A21 X11^T
        add({{t2, t2, k1, 0, true}, {t2, t1, k1, 2, false}});   // A22 -= L21 L21^T  and  L21 X11, one launch
    }
    p3_collect(pl, mid, hi);
    if (big) add({{(int)(n2 / 256), (int)(n1 / 256), (int)(n2 / 32), 3, false}});  // X21 = -X22 (L21 X11)
}
static std::shared_ptr<const P3Plans> p3_plans_for(int64_t nblk) {
    static std::mutex mu;
    static std::map<std::pair<int64_t, int64_t>, std::shared_ptr<const P3Plans>> cache;
    std::lock_guard<std::mutex> lk(mu);
    const std::pair<int64_t, int64_t> key{nblk, p3_min() * 4 + p3_planes()};
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    auto pl = std::make_shared<P3Plans>();
    p3_collect(*pl, 0, nblk);
    cache[key] = pl;
    return pl;
}
struct P3Run {  // per gq_h_prepare call
    const P3Plans* plans = nullptr;
    const uint32_t* words_dev = nullptr;
    size_t next = 0;
    unsigned char* img[2] = {nullptr, nullptr};
    float* partial = nullptr;
    unsigned* rmax = nullptr;     // NP == 2: [2][n/2] row maxima (bits) ...
    float* inv_scale = nullptr;   // ... and [2][n/2] epilogue scales
};
static size_t p3_image_bytes(int64_t C) { return (size_t)(C / 2) * (size_t)(C / 2) * 2 * (size_t)p3_planes(); }
static bool p3_used(int64_t C) {
    const int64_t nblk = C / NB, mid = nblk / 2;
    return nblk >= 2 && p3_node(mid * NB, (nblk - mid) * NB);
}

size_t h_prepare_workspace_bytes(int64_t R, int64_t C) {
    (void)R;
    const size_t n2 = (size_t)C * (size_t)C * sizeof(float);
    size_t extra = 0;
    if (p3_used(C)) {
        auto pl = p3_plans_for(C / NB);
        extra = 2 * (p3_image_bytes(C) + 256) + (size_t)P3_MAX_SLOTS * p3::TILE * p3::TILE * 4 + pl->words.size() * 4 +
                4 * (size_t)C * 4 + 2048;
    }
    return 2 * n2 + 2 * (size_t)C + 4 * (size_t)C + 1024 + extra;
}

// Recursive blocked Cholesky WITH inverse, all level-3 work on the fp32 matrix cores:
//   [A11 .  ]      L11 = chol(A11), X11 = L11^-1                     (recursion)
//   [A21 A22]      L21 = A21 X11^T                                   (GEMM, X11 triangular: k-range skip)
//                  A22 -= L21 L21^T                                  (SYRK, lower tiles only)
//                  L22 = chol(A22), X22 = L22^-1                     (recursion)
//                  X21 = -X22 (L21 X11)                              (two GEMMs, triangular k-range skips)
// L21 lives in Tmp (the factor itself is not an output: only X = L^-1 is), the product L21 X11
// reuses the dead A21 block.  Leaves are 128x128 (diag_potrf_inv_kernel, writes X_kk directly).
// (Measured and dropped, r02: L21 X11 of the large nodes on helper streams -- one per recursion depth -- under the
// recursion into A22, which it does not depend on: h_prepare(14336) alone 24.3 -> 23.5 ms, but inside a block's
// four-chain schedule the three extra hardware queues cost far more than that: 102 -> 113 ms per step.)
template <int NP>
static p3::Problem p3_problem(P3Run& run, const unsigned char* Aimg, int64_t Ka, const unsigned char* Bimg, int64_t Kb,
                              float* Cm, int64_t ldc, int mode, const float* rs, const float* cs) {
    return p3::Problem{Aimg, Bimg, (uint32_t)(Ka / 32 * NP * p3::BLK), (uint32_t)(Kb / 32 * NP * p3::BLK), Cm, ldc, mode,
                       rs, cs, run.partial};
}
template <int NP>
static int p3_gemm(P3Run& run, const p3::Problem& p0, const p3::Problem* p1, hipStream_t st) {
    const P3Gemm& gm = run.plans->gemms[run.next++];
    p3::Group g;
    g.p[0] = p0;
    g.p[1] = p1 ? *p1 : p0;
    g.table = run.words_dev + gm.table_at;
    return p3::launch_gemm<NP>(g, gm.n_reduce, run.words_dev + gm.rlist_at, st);
}

static int chol_inv_rec(float* A, float* X, float* Tmp, int* flag, int64_t n, int64_t lo, int64_t hi, size_t diag_lds,
                        hipStream_t st, P3Run& run);

// One node of the recursion on the image GEMMs.  Same products as the generic path below; L21 X11 moves in front of
// the recursion into A22 (it does not depend on it) and shares the launch -- and the image of L21 -- with the SYRK update.
template <int NP>
static int chol_node_p3(float* A, float* X, float* Tmp, int* flag, int64_t n, int64_t lo, int64_t mid, int64_t hi,
                        size_t diag_lds, hipStream_t st, P3Run& run) {
    const int64_t n1 = (mid - lo) * NB, n2 = (hi - mid) * NB;
    const int64_t o21 = (mid * NB) * n + lo * NB, o11 = (lo * NB) * n + lo * NB, o22 = (mid * NB) * n + mid * NB;
    unsigned* rm0 = run.rmax, *rm1 = run.rmax ? run.rmax + n / 2 : nullptr;
    float* is0 = run.inv_scale, *is1 = run.inv_scale ? run.inv_scale + n / 2 : nullptr;
    const float* s0 = NP == 2 ? is0 : nullptr;
    const float* s1 = NP == 2 ? is1 : nullptr;
    int rc;
    {
        ProfScope ps(PT_CHOL_GEMM, st);
        // L21 = A21 X11^T  (X11 lower-triangular: k < 256 (tn + 1))
        if ((rc = p3::launch_split<NP>(false, A + o21, n, n2, n1, 0, 0, run.img[0], rm0, is0, st))) return rc;
        if ((rc = p3::launch_split<NP>(false, X + o11, n, n1, n1, 1, 0, run.img[1], rm1, is1, st))) return rc;
        const p3::Problem g1 = p3_problem<NP>(run, run.img[0], n1, run.img[1], n1, Tmp + o21, n, 1, s0, s1);
        if ((rc = p3_gemm<NP>(run, g1, nullptr, st))) return rc;
        // A22 -= L21 L21^T (lower tiles) and A21 <- L21 X11 in ONE launch: both need only L21.  X11[k][j] = 0 for k < j,
        // so both images have their chunks in reverse order: every tile of the second product starts at chunk 0
        if ((rc = p3::launch_split<NP>(false, Tmp + o21, n, n2, n1, 0, 1, run.img[0], rm0, is0, st))) return rc;
        if ((rc = p3::launch_split<NP>(true, X + o11, n, n1, n1, 2, 1, run.img[1], rm1, is1, st))) return rc;
        const p3::Problem g2 = p3_problem<NP>(run, run.img[0], n1, run.img[0], n1, A + o22, n, 0, s0, s0);
        const p3::Problem g3 = p3_problem<NP>(run, run.img[0], n1, run.img[1], n1, A + o21, n, 1, s0, s1);
        if ((rc = p3_gemm<NP>(run, g2, &g3, st))) return rc;
    }
    if ((rc = chol_inv_rec(A, X, Tmp, flag, n, mid, hi, diag_lds, st, run))) return rc;
    ProfScope ps(PT_TRTRI_GEMM, st);
    // X21 = -X22 (L21 X11)  (X22 lower-triangular: k < 256 (tm + 1))
    if ((rc = p3::launch_split<NP>(false, X + o22, n, n2, n2, 1, 0, run.img[0], rm0, is0, st))) return rc;
    if ((rc = p3::launch_split<NP>(true, A + o21, n, n1, n2, 0, 0, run.img[1], rm1, is1, st))) return rc;
    const p3::Problem g4 = p3_problem<NP>(run, run.img[0], n2, run.img[1], n2, X + o21, n, 2, s0, s1);
    return p3_gemm<NP>(run, g4, nullptr, st);
}

static int chol_inv_rec(float* A, float* X, float* Tmp, int* flag, int64_t n, int64_t lo, int64_t hi, size_t diag_lds,
                        hipStream_t st, P3Run& run) {
    if (hi - lo == 1) {
        ProfScope ps(PT_DIAG_POTRF, st);
        const int64_t o = (lo * NB) * n + lo * NB;
        if (diag_lds == DIAG5_LDS)
            hipLaunchKernelGGL(diag_blk5_kernel, dim3(1), dim3(320), diag_lds, st, A + o, n, X + o, n, flag);
        else
            hipLaunchKernelGGL(diag_potrf_inv_kernel, dim3(1), dim3(256), diag_lds, st, A + o, n, X + o, n, flag);
        GQ_LAUNCH_CHECK();
        return GQ_OK;
    }
    const int64_t mid = (lo + hi) / 2;
    int rc;
    if ((rc = chol_inv_rec(A, X, Tmp, flag, n, lo, mid, diag_lds, st, run))) return rc;
    const int64_t n1 = (mid - lo) * NB, n2 = (hi - mid) * NB;
    if (run.plans && p3_node(n1, n2))
        return p3_planes() == 3 ? chol_node_p3<3>(A, X, Tmp, flag, n, lo, mid, hi, diag_lds, st, run)
                                : chol_node_p3<2>(A, X, Tmp, flag, n, lo, mid, hi, diag_lds, st, run);
    const int64_t o21 = (mid * NB) * n + lo * NB, o11 = (lo * NB) * n + lo * NB, o22 = (mid * NB) * n + mid * NB;
    // large nodes: fp32-accurate products on the bf16 matrix cores (gq_gemm3b.hpp); small ones are
    // latency-bound and stay on the fp32 instruction
    const int64_t min3b = opt(OPT_chol_fp32) ? (int64_t)1 << 40 : opt(OPT_chol_3b_min);
    const bool big = n1 >= min3b && n2 >= min3b;
#define GQ_CHOL_GEMM(TB, MODE, LOW, KRV, ...) \
    (big ? launch_gemm3b<TB, MODE, LOW, KRV>(__VA_ARGS__) : launch_gemm32<TB, MODE, LOW, KRV>(__VA_ARGS__))
    // small nodes: the SYRK update and L21 X11 (both need only L21) share one launch of whole 64-tiles, in front of the
    // recursion into A22 (bit-identical to separate launches: every output element is the same k-ordered chain)
    const bool pair = !big && !opt(OPT_chol_no_pair) && n % 4 == 0 && gemm32_uses_64_full(n2, n2, n1, true) &&
                      gemm32_uses_64_full(n2, n1, n1, false);
    {
        ProfScope ps(PT_CHOL_GEMM, st);
        if ((rc = GQ_CHOL_GEMM(true, 1, false, 1, Tmp + o21, n, A + o21, n, X + o11, n, n2, n1, n1, st))) return rc;
        if (pair) {
            if ((rc = launch_gemm32_pair(A + o22, Tmp + o21, n2, n1, A + o21, Tmp + o21, X + o11, n2, n1, n1, n, st))) return rc;
        } else if ((rc = GQ_CHOL_GEMM(true, 0, true, 0, A + o22, n, Tmp + o21, n, Tmp + o21, n, n2, n2, n1, st))) return rc;
    }
    if ((rc = chol_inv_rec(A, X, Tmp, flag, n, mid, hi, diag_lds, st, run))) return rc;
    ProfScope ps(PT_TRTRI_GEMM, st);
    if (!pair && (rc = GQ_CHOL_GEMM(false, 1, false, 2, A + o21, n, Tmp + o21, n, X + o11, n, n2, n1, n1, st))) return rc;
    return GQ_CHOL_GEMM(false, 2, false, 3, X + o21, n, X + o22, n, A + o21, n, n2, n1, n2, st);
#undef GQ_CHOL_GEMM
}

int w_prepare(const uint8_t* flags, float* W, int64_t R, int64_t C, int* mismatch, hipStream_t st) {
    if (!flags || !W || !mismatch) GQ_FAIL(GQ_E_NULL, "gq_w_prepare: null pointer");
    if (R <= 0 || C <= 0) GQ_FAIL(GQ_E_BAD_SHAPE, "gq_w_prepare: R=%ld C=%ld", (long)R, (long)C);
    GQ_HIP(hipMemsetAsync(mismatch, 0, sizeof(int), st));
    ProfScope ps(PT_PREP_ELEM, st);
    if (C % 4 == 0 && (uintptr_t)W % 16 == 0)
        hipLaunchKernelGGL(w_prepare4_kernel, dim3((unsigned)((C + 63) / 64)), dim3(256), 0, st, W, R, C, flags, flags + C,
                           mismatch);
    else
        hipLaunchKernelGGL(w_prepare_kernel, dim3((unsigned)((C + 63) / 64)), dim3(256), 0, st, W, R, C, flags, flags + C,
                           mismatch);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

// obq_order: EvoPress FastOBQ (evopress/src/fast_obq.py:133-141, 221-228) fixes the dead diagonal and damps FIRST,
// then masks the zero columns of W with an undamped 1 on the diagonal; GPTQ (gptq.py:308-316) masks, then damps.
int h_prepare(float* H, float* W, int64_t R, int64_t C, float rel_damp, float* U, int* not_invertible,
              uint8_t* col_flags_out, void* ws, size_t ws_bytes, hipStream_t st, bool obq_order) {
    if (!H || !W || !U || !not_invertible) GQ_FAIL(GQ_E_NULL, "gq_h_prepare: null pointer");
    if (R <= 0 || C <= 0 || C % NB) GQ_FAIL(GQ_E_BAD_SHAPE, "gq_h_prepare: R=%ld C=%ld (C %% 128 != 0)", (long)R, (long)C);
    const size_t need = h_prepare_workspace_bytes(R, C);
    if (!ws || ws_bytes < need) GQ_FAIL(GQ_E_WORKSPACE, "gq_h_prepare: workspace %zu < %zu bytes", ws_bytes, need);
    const int64_t n = C, nblk = C / NB;
    float* A = reinterpret_cast<float*>(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    float* X = A + (size_t)n * n;
    uint8_t* dead = reinterpret_cast<uint8_t*>(X + (size_t)n * n);
    uint8_t* zc = dead + n;
    int rc;

    P3Run run;
    std::shared_ptr<const P3Plans> plans;
    if (p3_used(C)) {
        plans = p3_plans_for(nblk);
        unsigned char* p = reinterpret_cast<unsigned char*>(zc + n);
        p = reinterpret_cast<unsigned char*>(((uintptr_t)p + 255) & ~(uintptr_t)255) + (((size_t)n * 4 + 255) & ~(size_t)255);
        run.img[0] = p; p += (p3_image_bytes(C) + 255) & ~(size_t)255;
        run.img[1] = p; p += (p3_image_bytes(C) + 255) & ~(size_t)255;
        run.partial = reinterpret_cast<float*>(p); p += (size_t)P3_MAX_SLOTS * p3::TILE * p3::TILE * 4;
        if (p3_planes() == 2) {
            run.rmax = reinterpret_cast<unsigned*>(p); p += (size_t)n * 4;
            run.inv_scale = reinterpret_cast<float*>(p); p += (size_t)n * 4;
        } else {
            p += 2 * (size_t)n * 4;
        }
        p = reinterpret_cast<unsigned char*>(((uintptr_t)p + 255) & ~(uintptr_t)255);
        if (p + plans->words.size() * 4 > reinterpret_cast<unsigned char*>(ws) + ws_bytes)
            GQ_FAIL(GQ_E_WORKSPACE, "gq_h_prepare: workspace too small for the image GEMMs");
        // pageable source: staged before the call returns; the plans object outlives it (cached for the process)
        GQ_HIP(hipMemcpyAsync(p, plans->words.data(), plans->words.size() * 4, hipMemcpyHostToDevice, st));
        run.words_dev = reinterpret_cast<const uint32_t*>(p);
        run.plans = plans.get();
    }
    GQ_HIP(hipMemsetAsync(not_invertible, 0, sizeof(int), st));
    float* eq_s = nullptr;
    {
    ProfScope ps(PT_PREP_ELEM, st);
    if ((uintptr_t)W % 16 == 0)  // C % 128 == 0 here
        hipLaunchKernelGGL(col_flags4_kernel, dim3((unsigned)((C + 63) / 64)), dim3(256), 0, st, H, W, R, C, dead, zc);
    else
        hipLaunchKernelGGL(col_flags_kernel, dim3((unsigned)((C + 63) / 64)), dim3(256), 0, st, H, W, R, C, dead, zc);
    GQ_LAUNCH_CHECK();
    hipLaunchKernelGGL(zero_dead_cols_kernel, dim3((unsigned)((C + 63) / 64)), dim3(256), 0, st, W, R, C, dead);
    GQ_LAUNCH_CHECK();
    if (col_flags_out) GQ_HIP(hipMemcpyAsync(col_flags_out, dead, 2 * (size_t)C, hipMemcpyDeviceToDevice, st));
    if (obq_order) {
        hipLaunchKernelGGL(dead_diag_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, st, H, C, dead);
        hipLaunchKernelGGL(damp_kernel, dim3(1), dim3(1024), 0, st, H, C, rel_damp);
        hipLaunchKernelGGL(mask_h_kernel, dim3((unsigned)((C + 63) / 64)), dim3(256), 0, st, H, C, zc);
    } else {
        hipLaunchKernelGGL(mask_h_kernel, dim3((unsigned)((C + 63) / 64)), dim3(256), 0, st, H, C, zc);
        hipLaunchKernelGGL(damp_kernel, dim3(1), dim3(1024), 0, st, H, C, rel_damp);
    }
    GQ_LAUNCH_CHECK();
    const bool equil = !opt(OPT_chol_no_equil);
    if (equil) {
        eq_s = reinterpret_cast<float*>(((uintptr_t)(zc + n) + 255) & ~(uintptr_t)255);
        hipLaunchKernelGGL(equil_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, H, n, eq_s);
    }
    // Neither A above its block diagonal nor X needs clearing: every block that is read is written first (diagonal
    // blocks whole, zeros included).  The test that pins this fills both with NaN patterns first (option chol_poison) and
    // expects the same U.
    if (opt(OPT_chol_poison)) GQ_HIP(hipMemsetAsync(A, 0xff, 2 * (size_t)n * n * sizeof(float), st));
    hipLaunchKernelGGL(reverse_copy_kernel, dim3((unsigned)n), dim3(256), 0, st, A, H, n, eq_s);
    GQ_LAUNCH_CHECK();
    }
    static std::atomic<bool> attr_set{false};  // guards an idempotent call: a race sets the same value twice
    const bool use_ref = opt(OPT_diag_ref) != 0;  // the column-by-column kernel
    const size_t diag_lds = use_ref ? (2 * NB * LDP + NB) * sizeof(float) : DIAG5_LDS;
    if (!attr_set) {
        GQ_HIP(hipFuncSetAttribute((const void*)diag_potrf_inv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)((2 * NB * LDP + NB) * sizeof(float))));
        GQ_HIP(hipFuncSetAttribute((const void*)diag_blk5_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)DIAG5_LDS));
        attr_set = true;
    }
    if ((rc = chol_inv_rec(A, X, U, not_invertible, n, 0, nblk, diag_lds, st, run))) return rc;
    ProfScope ps(PT_PREP_ELEM, st);
    hipLaunchKernelGGL(finish_u_kernel, dim3((unsigned)n), dim3(256), 0, st, U, X, n, not_invertible, eq_s);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

// ---- gq_chol_gemm: one product of the chain through the image path, for tests and benchmarks ----
size_t chol_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    const size_t img = (size_t)3 * 2 * (size_t)K;
    return img * (size_t)M + img * (size_t)N + (size_t)P3_MAX_SLOTS * p3::TILE * p3::TILE * 4 + 2 * (size_t)(M + N) * 4 +
           ((size_t)(M / 256 + 1) * (size_t)(N / 256 + 1) * 6 * 4 + 64) * 4 + 4096;
}
template <int NP>
static int chol_gemm_np(float* Cm, int64_t ldc, const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M, int64_t N,
                        int64_t K, int trans_b, int mode, int kr, int lower, unsigned char* p, unsigned char* end,
                        hipStream_t st) {
    auto take = [&](size_t bytes) {
        unsigned char* q = p;
        p += (bytes + 255) & ~(size_t)255;
        return q;
    };
    unsigned char* ia = take((size_t)NP * 2 * M * K);
    unsigned char* ib = take((size_t)NP * 2 * N * K);
    float* partial = reinterpret_cast<float*>(take((size_t)P3_MAX_SLOTS * p3::TILE * p3::TILE * 4));
    unsigned* rma = reinterpret_cast<unsigned*>(take((size_t)M * 4));
    unsigned* rmb = reinterpret_cast<unsigned*>(take((size_t)N * 4));
    float* isa = reinterpret_cast<float*>(take((size_t)M * 4));
    float* isb = reinterpret_cast<float*>(take((size_t)N * 4));
    const p3::GemmShape sh{(int)(M / 256), (int)(N / 256), (int)(K / 32), kr, lower != 0};
    const p3::Plan pl = p3::make_plan({sh}, NP == 3 ? 2 : 4, P3_MAX_SLOTS);
    if (pl.nslots < 0) GQ_FAIL(GQ_E_UNSUPPORTED, "gq_chol_gemm: the schedule needs more than %d partial slots", P3_MAX_SLOTS);
    std::vector<uint32_t> words(pl.table);
    const size_t rl_at = words.size();
    words.insert(words.end(), pl.rlist.begin(), pl.rlist.end());
    uint32_t* wd = reinterpret_cast<uint32_t*>(take(words.size() * 4));
    if (p > end) GQ_FAIL(GQ_E_WORKSPACE, "gq_chol_gemm: workspace too small");
    GQ_HIP(hipMemcpyAsync(wd, words.data(), words.size() * 4, hipMemcpyHostToDevice, st));  // pageable: staged before return
    const int rev = kr == 2;
    int rc;
    if ((rc = p3::launch_split<NP>(false, A, lda, M, K, kr == 3 ? 1 : 0, rev, ia, rma, isa, st))) return rc;
    if (!(lower && A == B && trans_b)) {
        if ((rc = p3::launch_split<NP>(!trans_b, B, ldb, N, K, kr == 1 ? 1 : (kr == 2 ? 2 : 0), rev, ib, rmb, isb, st))) return rc;
    } else {
        ib = ia;
        isb = isa;
    }
    p3::Group g;
    g.p[0] = p3::Problem{ia, ib, (uint32_t)(K / 32 * NP * p3::BLK), (uint32_t)(K / 32 * NP * p3::BLK), Cm, ldc, mode,
                         NP == 2 ? isa : nullptr, NP == 2 ? isb : nullptr, partial};
    g.p[1] = g.p[0];
    g.table = wd;
    return p3::launch_gemm<NP>(g, (int)(pl.rlist.size() / 2), wd + rl_at, st);
}
int chol_gemm(float* Cm, int64_t ldc, const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M, int64_t N, int64_t K,
              int trans_b, int mode, int kr, int lower, int planes, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!Cm || !A || !B || !ws) GQ_FAIL(GQ_E_NULL, "gq_chol_gemm: null pointer");
    if (M <= 0 || N <= 0 || K <= 0 || M % 256 || N % 256 || K % 128 || lda % 4 || ldb % 4 || ldc % 4)
        GQ_FAIL(GQ_E_BAD_SHAPE, "gq_chol_gemm: M=%ld N=%ld (multiples of 256) K=%ld (of 128), ld %% 4 == 0", (long)M, (long)N, (long)K);
    if (mode < 0 || mode > 2 || kr < 0 || kr > 3 || (planes != 2 && planes != 3) || (kr == 1 && !trans_b) || (kr == 2 && trans_b))
        GQ_FAIL(GQ_E_UNSUPPORTED, "gq_chol_gemm: mode=%d kr=%d planes=%d trans_b=%d", mode, kr, planes, trans_b);
    unsigned char* p = reinterpret_cast<unsigned char*>(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    unsigned char* end = reinterpret_cast<unsigned char*>(ws) + ws_bytes;
    return planes == 3 ? chol_gemm_np<3>(Cm, ldc, A, lda, B, ldb, M, N, K, trans_b, mode, kr, lower, p, end, st)
                       : chol_gemm_np<2>(Cm, ldc, A, lda, B, ldb, M, N, K, trans_b, mode, kr, lower, p, end, st);
}

}  // namespace gq
