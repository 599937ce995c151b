// gq_gemm32.hpp -- fp32 GEMM on v_mfma_f32_32x32x2_f32 (exact fp32: every output is a
// k-ordered fmaf chain), shared by the GPTQ trailing update (K6) and the blocked
// Cholesky / triangular inverse (K3).
//
//   MODE 0:  C = C - A op(B)      (acc chain starts at 0, then ONE subtraction)
//   MODE 1:  C = A op(B)
//   MODE 2:  C = -(A op(B))
//   TRANS_B: op(B) = B^T with B stored [N,K] row-major (else B is [K,N] row-major)
//   LOWER:   only tiles with tile_row >= tile_col are computed (square, symmetric updates)
//   KR:      k-range restriction for triangular operands (zeros are skipped, not multiplied):
//            0 full; 1: k < n0+128 (B^T lower-triangular); 2: k >= n0 (B lower-triangular);
//            3: k < m0+128 (A lower-triangular)
//   CHAIN:   (MODE 0 only) the accumulation chain restarts every CHAIN k:  C = (..((C - A_0 B_0) - A_1 B_1)..)
//            with A_i B_i the i-th CHAIN-wide slice of the product -- bit-identical to K/CHAIN separate
//            MODE-0 launches, with ONE read and write of C (the GPTQ look-ahead trailing update)
//
// Workgroup = 256 threads = 4 waves (2x2); tile TS x TS, TS = 128 (each wave owns 64x64 = 2x2 MFMA tiles,
// 64 accumulator VGPRs) or TS = 64 (one MFMA tile per wave) for problems too small to fill the chip with
// 128-tiles: a 128x128x128 tile is 8.6 us of matrix-pipe time on ONE CU, four 64-tiles are 2.1 us on four.
// K streams in chunks of 32 through a double-buffered LDS image: chunk t+1 is written to the other image in
// the middle of the MFMA block of chunk t and the loads of chunk t+2 follow it (one barrier per chunk, one
// basic block per chunk).  FULL kernels (whole tiles) load without predicates.  The CHAIN kernel also holds
// the C tile and runs 8 waves (2x4, wave tile 64x32); the GPTQ far update on whole tiles has its own kernel,
// gemm32_chain_full_kernel below.
#pragma once
#include <atomic>
#include <stdlib.h>

#include <type_traits>

#include "gq_common.hpp"

namespace gq {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int TM = 128, TN = 128, TK = 32;  // the default tile (gq_gemm3b.hpp uses these too)
constexpr int LDA_S = TK + 1;   // A tile [TS][TK]: lanes walk rows -> odd stride
constexpr int LDBT_S = TK + 1;  // B tile [TS][TK] (NT)
template <int TS> struct G32 {
    static constexpr int NV = TS / 32;       // float4 per thread and operand chunk
    static constexpr int LDB = TS + 4;       // B tile [TK][TS] (NN): lanes walk columns
    static constexpr int A_FLOATS = TS * LDA_S;
    static constexpr int B_FLOATS = (TK * LDB > TS * LDBT_S) ? TK * LDB : TS * LDBT_S;
    static constexpr int STAGE_FLOATS = A_FLOATS + B_FLOATS;
    static constexpr int LDS_BYTES = 2 * STAGE_FLOATS * 4;
};
constexpr int LDB_S = G32<128>::LDB;
constexpr int G32_LDS_BYTES = G32<128>::LDS_BYTES;

// rows x 32-float panel chunk -> NV float4 per thread (32 NV rows x 8 float4)
template <int NV, int NT = 256>
__device__ __forceinline__ void g32_load_rows(float4 (&v)[NV], const float* P, int64_t ld, int64_t r0, int64_t rmax,
                                              int64_t k0, int64_t K, int tid) {
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        const int idx = tid + t * NT, rr = idx >> 3, c4 = (idx & 7) * 4;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r0 + rr < rmax) {
            const float* p = P + (r0 + rr) * ld + k0 + c4;
            if (k0 + c4 + 3 < K) x = *reinterpret_cast<const float4*>(p);
            else {
                if (k0 + c4 + 0 < K) x.x = p[0];
                if (k0 + c4 + 1 < K) x.y = p[1];
                if (k0 + c4 + 2 < K) x.z = p[2];
            }
        }
        v[t] = x;
    }
}
template <int NV, int NT = 256>
__device__ __forceinline__ void g32_store_rows(const float4 (&v)[NV], float* S, int lds, int tid) {
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        const int idx = tid + t * NT, rr = idx >> 3, c4 = (idx & 7) * 4;
        float* o = S + rr * lds + c4;
        o[0] = v[t].x; o[1] = v[t].y; o[2] = v[t].z; o[3] = v[t].w;
    }
}
// 32 k-rows x TS columns chunk of a [K,N] matrix -> NV float4 per thread (32 rows x TS/4 float4)
template <int NV, int NT = 256>
__device__ __forceinline__ void g32_load_kn(float4 (&v)[NV], const float* B, int64_t ldb, int64_t n0, int64_t N,
                                            int64_t k0, int64_t K, int tid) {
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        const int idx = tid + t * NT, kk = idx / (NV * NT / 32), c4 = (idx % (NV * NT / 32)) * 4;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k0 + kk < K) {
            const float* p = B + (k0 + kk) * ldb + n0 + c4;
            if (n0 + c4 + 3 < N) x = *reinterpret_cast<const float4*>(p);
            else {
                if (n0 + c4 + 0 < N) x.x = p[0];
                if (n0 + c4 + 1 < N) x.y = p[1];
                if (n0 + c4 + 2 < N) x.z = p[2];
            }
        }
        v[t] = x;
    }
}
template <int NV, int NT = 256>
__device__ __forceinline__ void g32_store_kn(const float4 (&v)[NV], float* S, int tid) {
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        const int idx = tid + t * NT, kk = idx / (NV * NT / 32), c4 = (idx % (NV * NT / 32)) * 4;
        *reinterpret_cast<float4*>(S + kk * (NV * NT / 8 + 4) + c4) = v[t];
    }
}

// whole-tile variants (FULL kernels: M, N multiples of TS and K of TK): plain 16-byte loads, no predicates
template <int NV, int NT = 256>
__device__ __forceinline__ void g32_load_rows_full(float4 (&v)[NV], const float* P, int64_t ld, int64_t r0, int64_t k0,
                                                   int tid) {
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        const int idx = tid + t * NT, rr = idx >> 3, c4 = (idx & 7) * 4;
        v[t] = *reinterpret_cast<const float4*>(P + (r0 + rr) * ld + k0 + c4);
    }
}
template <int NV, int NT = 256>
__device__ __forceinline__ void g32_load_kn_full(float4 (&v)[NV], const float* B, int64_t ldb, int64_t n0, int64_t k0,
                                                 int tid) {
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        const int idx = tid + t * NT, kk = idx / (NV * NT / 32), c4 = (idx % (NV * NT / 32)) * 4;
        const float4 x = *reinterpret_cast<const float4*>(B + (k0 + kk) * ldb + n0 + c4);
        v[t].x = x.x; v[t].y = x.y; v[t].z = x.z; v[t].w = x.w;  // member-wise: a whole-vector copy keeps v[] in scratch
    }
}

constexpr int G32_COMMIT_AT = 24;  // k position (of 32) where chunk t+1 is written to LDS: 3/4 through the MFMA block
// one output tile (tile column bxr of gx, tile row byr of gy, in dispatch order) by the calling workgroup
template <bool TRANS_B, int MODE, bool LOWER, int KR = 0, int CHAIN = 0, int TS = 128, bool FULL = false, int NW = 4>
__device__ __forceinline__ void gemm32_tile(float* Cmat, int64_t ldc, const float* A, int64_t lda, const float* B,
                                            int64_t ldb, int64_t M, int64_t N, int64_t K, unsigned bxr, unsigned byr,
                                            unsigned gx, unsigned gy) {
    extern __shared__ __attribute__((aligned(16))) float g32_smem[];
    // triangular k-ranges make tile cost grow with n0 (KR 1) or m0 (KR 3): dispatch the long tiles first
    const unsigned bx = (KR == 1) ? gx - 1 - bxr : bxr;
    const unsigned by = (KR == 3) ? gy - 1 - byr : byr;
    if (LOWER && bx > by) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // waves 2 x NW/2: wave tile WTM x WTN = NIM x NIN MFMA tiles (NW = 8 halves the accumulator registers
    // of the CHAIN kernel, which also holds the C tile, so that two workgroups = 16 waves share a CU)
    constexpr int NT = NW * 64, WNW = NW / 2;
    const int wm = wid / WNW, wn = wid % WNW;
    constexpr int WTM = TS / 2, WTN = TS / WNW, NIM = WTM / 32, NIN = WTN / 32, NV = TS * 8 / NT;
    static_assert(NIN >= 1 && NV >= 1, "tile too small for this many waves");
    const int64_t m0 = (int64_t)by * TS, n0 = (int64_t)bx * TS;
    f32x16 acc[NIM][NIN];
#pragma unroll
    for (int i = 0; i < NIM; ++i)
#pragma unroll
        for (int j = 0; j < NIN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    static_assert(CHAIN == 0 || (MODE == 0 && CHAIN % TK == 0 && KR == 0), "CHAIN: MODE 0, whole stages");
    const int lc = lane & 31, lh = lane >> 5;
    f32x16 cv[CHAIN ? NIM : 1][CHAIN ? NIN : 1];
    if constexpr (CHAIN != 0) {
#pragma unroll
        for (int i = 0; i < NIM; ++i)
#pragma unroll
            for (int j = 0; j < NIN; ++j) {
                const int64_t col = n0 + wn * WTN + j * 32 + lc;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t rowi = m0 + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    cv[i][j][e] = (FULL || (rowi < M && col < N)) ? Cmat[rowi * ldc + col] : 0.f;
                }
            }
    }
    float4 va[NV], vb[NV];
    auto fetch = [&](int64_t k0) {
        if constexpr (FULL) {
            g32_load_rows_full<NV, NT>(va, A, lda, m0, k0, tid);
            if constexpr (TRANS_B) g32_load_rows_full<NV, NT>(vb, B, ldb, n0, k0, tid);
            else g32_load_kn_full<NV, NT>(vb, B, ldb, n0, k0, tid);
        } else {
            g32_load_rows<NV, NT>(va, A, lda, m0, M, k0, K, tid);
            if constexpr (TRANS_B) g32_load_rows<NV, NT>(vb, B, ldb, n0, N, k0, K, tid);
            else g32_load_kn<NV, NT>(vb, B, ldb, n0, N, k0, K, tid);
        }
    };
    auto commit = [&](int buf) {
        float* As = g32_smem + buf * G32<TS>::STAGE_FLOATS;
        float* Bs = As + G32<TS>::A_FLOATS;
        g32_store_rows<NV, NT>(va, As, LDA_S, tid);
        if constexpr (TRANS_B) g32_store_rows<NV, NT>(vb, Bs, LDBT_S, tid);
        else g32_store_kn<NV, NT>(vb, Bs, tid);
    };
    const int li = lane & 31, lk = lane >> 5;
    int64_t kb = 0, ke = K;
    if constexpr (KR == 1) ke = (n0 + TS < K) ? n0 + TS : K;
    if constexpr (KR == 2) kb = (n0 < K) ? n0 : K;
    if constexpr (KR == 3) ke = (m0 + TS < K) ? m0 + TS : K;
    const int64_t nk = (ke - kb + TK - 1) / TK;
    // Software pipeline: chunk t+1 is written to the other LDS buffer in the MIDDLE of the MFMA block of
    // chunk t (its loads were issued half a chunk + one barrier earlier) and the loads of chunk t+2 follow
    // it, so the LDS writes, the address arithmetic and the global loads all issue in the shadow of MFMAs.
    if (nk > 0) {
        fetch(kb);
        commit(0);
        fetch(kb + ((nk > 1) ? TK : 0));
    }
    __syncthreads();
    for (int64_t t = 0; t < nk; ++t) {
        const float* As = g32_smem + (t & 1) * G32<TS>::STAGE_FLOATS;
        const float* Bs = As + G32<TS>::A_FLOATS;
        // operands of k-step kk+2 are read while the 4 MFMAs of step kk run (explicit register double buffer:
        // left to the compiler, every 4 MFMAs waited for their own LDS reads)
        float av[2][NIM], bv[2][NIN];
        auto frag = [&](int kk, float (&a)[NIM], float (&b)[NIN]) {
#pragma unroll
            for (int i = 0; i < NIM; ++i) a[i] = As[(wm * WTM + i * 32 + li) * LDA_S + kk + lk];
#pragma unroll
            for (int j = 0; j < NIN; ++j) {
                if constexpr (TRANS_B) b[j] = Bs[(wn * WTN + j * 32 + li) * LDBT_S + kk + lk];
                else b[j] = Bs[(kk + lk) * G32<TS>::LDB + wn * WTN + j * 32 + li];
            }
        };
        frag(0, av[0], bv[0]);
#pragma unroll
        for (int kk = 0; kk < TK; kk += 2) {
            const int cur = (kk >> 1) & 1;
            if (kk + 2 < TK) frag(kk + 2, av[cur ^ 1], bv[cur ^ 1]);
            if (kk == G32_COMMIT_AT) {
                // unconditional, so that the chunk body stays ONE basic block the scheduler can interleave:
                // past the end the last chunk is fetched again and committed to the buffer nobody reads
                commit((int)((t + 1) & 1));  // the other buffer: its readers passed the last barrier
                fetch(kb + ((t + 2 < nk) ? t + 2 : nk - 1) * TK);
            }
#pragma unroll
            for (int i = 0; i < NIM; ++i)
#pragma unroll
                for (int j = 0; j < NIN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i], bv[cur][j], acc[i][j], 0, 0, 0);
        }
        if constexpr (CHAIN != 0) {
            if (((t + 1) * TK) % CHAIN == 0) {  // end of a slice: one subtraction, new chain
#pragma unroll
                for (int i = 0; i < NIM; ++i)
#pragma unroll
                    for (int j = 0; j < NIN; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            cv[i][j][e] = cv[i][j][e] - acc[i][j][e];
                            acc[i][j][e] = 0.0f;
                        }
            }
        }
        __syncthreads();
    }
    // D layout: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < NIM; ++i)
#pragma unroll
        for (int j = 0; j < NIN; ++j) {
            const int64_t col = n0 + wn * WTN + j * 32 + lc;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t rowi = m0 + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (FULL || (rowi < M && col < N)) {
                    float* p = Cmat + rowi * ldc + col;
                    if constexpr (CHAIN != 0) *p = cv[i][j][e];
                    else if constexpr (MODE == 0) *p = *p - acc[i][j][e];
                    else if constexpr (MODE == 1) *p = acc[i][j][e];
                    else *p = -acc[i][j][e];
                }
            }
        }
}

template <bool TRANS_B, int MODE, bool LOWER, int KR = 0, int CHAIN = 0, int TS = 128, bool FULL = false, int NW = 4>
__global__ __launch_bounds__(NW * 64, 2) void gemm32_kernel(float* Cmat, int64_t ldc, const float* A, int64_t lda,
                                                        const float* B, int64_t ldb, int64_t M, int64_t N, int64_t K) {
    gemm32_tile<TRANS_B, MODE, LOWER, KR, CHAIN, TS, FULL, NW>(Cmat, ldc, A, lda, B, ldb, M, N, K, blockIdx.x, blockIdx.y,
                                                               gridDim.x, gridDim.y);
}

// Two independent products of one Cholesky recursion node in ONE launch (whole 64-tiles):
//   blocks [0, n_a):  C_a -= A_a A_a^T, lower tiles     (A22 -= L21 L21^T,  gemm32_kernel<true, 0, true, 0>)
//   the rest:         C_b  = A_b B_b,   k >= n0          (L21 X11,           gemm32_kernel<false, 1, false, 2>)
// Both need only L21; alone each is a latency-bound launch of a few microseconds on a fraction of the chip.
struct G32Pair {
    float* Ca; const float* Aa; int64_t Ma, Ka;          // SYRK update: [Ma, Ma] -= [Ma, Ka] [Ma, Ka]^T
    float* Cb; const float* Ab; const float* Bb; int64_t Mb, Nb, Kb;
    int64_t ld;                                          // one leading dimension: everything lives in n x n matrices
    unsigned gxa, n_a, gxb, gyb;                         // the SYRK's lower tiles are enumerated row by row
};
template <int TS>  // 64 (a template so that every translation unit including this header may instantiate it)
__global__ __launch_bounds__(256, 2) void gemm32_pair_kernel(const G32Pair p) {
    unsigned id = blockIdx.x;
    if (id < p.n_a) {
        // lower tile (by, bx), bx <= by, from the linear index id = by (by + 1) / 2 + bx
        unsigned by = (unsigned)((__fsqrt_rn(8.0f * (float)id + 1.0f) - 1.0f) * 0.5f);
        while ((by + 1) * (by + 2) / 2 <= id) ++by;
        while (by * (by + 1) / 2 > id) --by;
        const unsigned bx = id - by * (by + 1) / 2;
        gemm32_tile<true, 0, true, 0, 0, TS, true, 4>(p.Ca, p.ld, p.Aa, p.ld, p.Aa, p.ld, p.Ma, p.Ma, p.Ka, bx, by, p.gxa, p.gxa);
    } else {
        id -= p.n_a;
        gemm32_tile<false, 1, false, 2, 0, TS, true, 4>(p.Cb, p.ld, p.Ab, p.ld, p.Bb, p.ld, p.Mb, p.Nb, p.Kb, id % p.gxb,
                                                        id / p.gxb, p.gxb, p.gyb);
    }
}
inline int launch_gemm32_pair(float* Ca, const float* Aa, int64_t Ma, int64_t Ka, float* Cb, const float* Ab, const float* Bb,
                              int64_t Mb, int64_t Nb, int64_t Kb, int64_t ld, hipStream_t st) {
    static std::atomic<bool> attr_set{false};
    if (!attr_set) {
        GQ_HIP(hipFuncSetAttribute((const void*)gemm32_pair_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, G32<64>::LDS_BYTES));
        attr_set = true;
    }
    G32Pair p;
    p.Ca = Ca; p.Aa = Aa; p.Ma = Ma; p.Ka = Ka;
    p.Cb = Cb; p.Ab = Ab; p.Bb = Bb; p.Mb = Mb; p.Nb = Nb; p.Kb = Kb;
    p.ld = ld;
    p.gxa = (unsigned)(Ma / 64);
    p.n_a = p.gxa * (p.gxa + 1) / 2;
    p.gxb = (unsigned)(Nb / 64);
    p.gyb = (unsigned)(Mb / 64);
    hipLaunchKernelGGL(gemm32_pair_kernel<64>, dim3(p.n_a + p.gxb * p.gyb), dim3(256), G32<64>::LDS_BYTES, st, p);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}
// does launch_gemm32 pick whole 64-tiles for a Cholesky-node product of this size?
inline bool gemm32_uses_64_full(int64_t M, int64_t N, int64_t K, bool lower) {
    const int64_t max64 = opt(OPT_gemm32_64_max);
    const int64_t tiles128 = ((M + 127) / 128) * ((N + 127) / 128) / (lower ? 2 : 1);
    return tiles128 < max64 && M % 64 == 0 && N % 64 == 0 && K % TK == 0;
}

// ---- the GPTQ far trailing update on whole tiles: C = (..((C - A_0 B_0) - A_1 B_1)..), NN, 128x128 tiles ----
// Same arithmetic as gemm32_kernel<false, 0, false, 0, CHAIN, 128, true, 8> (every output element is the same
// sequence of k-ordered 128-long chains and subtractions), scheduled for the matrix pipe:
//   * fragment reads are inline asm with counted s_waitcnt: the compiler otherwise sinks every ds_read next
//     to its MFMA (measured: 74 % MFMA busy, each pair of MFMAs waiting ~100 cycles for its operands, and for
//     the LDS writes of the next chunk queued in front of them).  A group = 2 k-steps = 2 ds_read2_b32 (A) +
//     2 ds_read_b32 (B) + 4 MFMAs; reads run two groups (8 MFMAs) ahead in four rotating register sets;
//   * ring of THREE LDS images and the barrier in the MIDDLE of a chunk (before group 6, whose prefetch is the
//     first read of the next image): chunk boundaries have no barrier.  A writer of image (t+1)%3 has passed
//     barrier t-1, which every wave reaches only after its last read of that image in chunk t-2;
//   * the loop body is [second half of chain c | subtraction | first half of chain c+1]: the 32 subtractions
//     of a chain end interleave with the first MFMAs of the next chain, which start from the inline constant 0.
// 8 waves (2x4), wave tile 64x32, one workgroup per CU (99 KiB of LDS, <= 256 VGPRs).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t g32_u32x4 __attribute__((ext_vector_type(4)));
#define GQ_C_RD2(dst, addr, o0, o1) \
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(dst) : "v"(addr), "n"(o0), "n"(o1))
#define GQ_C_RD1(dst, addr, off) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
// every LDS operation older than the N most recent has completed; the "+v" operands make the consumers of
// the group's registers depend on the wait
#define GQ_C_WAIT(N, s) \
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fa0[s]), "+v"(fa1[s]), "+v"(fb0[s]), "+v"(fb1[s]) : "n"(N))

// CG: the group behind which a chunk's commit + fetch block stands.  Cycle stamps (profiles/r04_far_stamps.txt): the two waves of
// a SIMD do not alternate -- the older one wins the matrix pipe, runs ~3 groups ahead and parks ~770 cycles per chunk at the
// barrier (behind group 5) until its partner arrives; a commit block behind group 3 puts the PARTNER's stores and loads -- no
// MFMAs -- exactly into that window.  Behind group 1: far alone 0.762 -> 0.775 of the fp32 peak in the bench (r03: group 3).
// STAMP (profiles/micro/far_stamps.hip only; the library never instantiates it): every wave of the workgroup accumulates
// the shader-clock cycles it spends (a) between arriving at the chunk's vmcnt + barrier and leaving it, (b) in the chunk's
// commit + fetch block, and writes {loop cycles, barrier cycles, commit cycles, chunks} to stamps[4 wave ..].
template <int CHAIN, bool BDMA, bool STAMP = false, int CG = 1>
__device__ __forceinline__ void g32_chain_full_tile(float* Cmat, int64_t ldc, const float* A, int64_t lda, const float* B,
                                                    int64_t ldb, int64_t K, const int64_t m0, const int64_t n0,
                                                    unsigned long long* stamps = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float g32_smem[];
    // a stamp drains the LDS counter (SMEM returns out of order with LDS: the counted waits below must never see one in flight)
    auto now = [&]() -> unsigned long long {
        unsigned long long t_;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
        return t_;
    };
    unsigned long long st_bar = 0, st_commit = 0, st_t0 = 0, st_a = 0;
    // BDMA: the B chunk ([32 k][128 n], 512-byte rows as they lie in memory) goes global -> LDS by global_load_lds (two 1 KiB
    // pieces per wave and chunk, issued two chunks ahead), no VGPR round trip and no ds_write; A keeps the register-staged
    // transposing commit.  Same fragments, same MFMA order: bit-identical.
    constexpr int TS = 128, NT = 512, NV = 2, STAGE = G32<TS>::STAGE_FLOATS, AF = G32<TS>::A_FLOATS, LDB = BDMA ? TS : G32<TS>::LDB;
    constexpr int SPC = CHAIN / TK;  // chunks per chain
    static_assert(SPC == 4 && TK == 32, "the loop body is written for 4 chunks of 32 k per chain");
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 2, wn = wid & 3;
    const int li = lane & 31, lk = lane >> 5;
    const int64_t nk = K / TK;
    f32x16 acc[2], cv[2];
    float* cp[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        cp[i] = Cmat + (m0 + wm * 64 + i * 32 + 4 * lk) * ldc + n0 + wn * 32 + li;
#pragma unroll
        for (int e = 0; e < 16; ++e) cv[i][e] = cp[i][((e & 3) + 8 * (e >> 2)) * ldc];
    }
    // global loads run TWO chunks ahead of the LDS write (measured: one chunk, ~2 us, does not cover the
    // latency tail under load): two register sets, chunk c lives in set c & 1
    float4 va[2][NV], vb[2][NV];
    auto fetch = [&](int64_t t, float4 (&a)[NV], float4 (&b)[NV]) {
        const int64_t k0 = ((t < nk) ? t : nk - 1) * TK;  // past the end: the last chunk again (never consumed)
        g32_load_rows_full<NV, NT>(a, A, lda, m0, k0, tid);
        if constexpr (!BDMA) g32_load_kn_full<NV, NT>(b, B, ldb, n0, k0, tid);
    };
    auto commit = [&](int buf, const float4 (&a)[NV], const float4 (&b)[NV]) {
        float* As = g32_smem + buf * STAGE;
        g32_store_rows<NV, NT>(a, As, LDA_S, tid);
        if constexpr (!BDMA) g32_store_kn<NV, NT>(b, As + AF, tid);
    };
    // LDS byte addresses of this lane's fragments in image 0 (the dynamic LDS segment starts at address 0
    // of the workgroup's allocation: no static __shared__ in this kernel)
    const unsigned lds0 = (unsigned)(uintptr_t)g32_smem;
    // BDMA: wave w brings k-rows 4 w .. 4 w + 3 of every B chunk: piece p = rows 4 w + 2 p (lanes 0-31) and + 1 (lanes 32-63)
    const int wu = __builtin_amdgcn_readfirstlane(wid);
    const unsigned vb0 = (unsigned)(((4 * wu + (lane >> 5)) * ldb + (lane & 31) * 4) * 4), vb1 = vb0 + (unsigned)(2 * ldb * 4);
    const unsigned ldsb = lds0 + (unsigned)(AF * 4 + 4 * wu * 512);
    auto dma_b = [&](int64_t t, int buf) {
        if constexpr (BDMA) {
            const int64_t k0 = ((t < nk) ? t : nk - 1) * TK;
            const float* src = B + k0 * ldb + n0;
            const unsigned d0 = ldsb + (unsigned)(buf * STAGE * 4);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(vb0), "s"(src), "s"(d0) : "memory");
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(vb1), "s"(src), "s"(d0 + 1024u) : "memory");
        }
    };
    const unsigned aoff0 = lds0 + ((wm * 64 + li) * LDA_S + lk) * 4, aoff1 = aoff0 + 32 * LDA_S * 4;
    const unsigned boff = lds0 + (AF + lk * LDB + wn * 32 + li) * 4;
    f32x2 fa0[4], fa1[4];  // [set]: A tile 0 / 1, k-steps (4g, 4g+2)
    float fb0[4], fb1[4];  // [set]: B at k-step 4g / 4g+2
    // reads of group g (0..7) of the image whose fragment bases are (pa0, pa1, pb) into set s
#define GQ_C_READS(g, s, pa0, pa1, pb)                  \
    do {                                                \
        GQ_C_RD2(fa0[s], pa0, 4 * (g), 4 * (g) + 2);    \
        GQ_C_RD2(fa1[s], pa1, 4 * (g), 4 * (g) + 2);    \
        GQ_C_RD1(fb0[s], pb, (4 * (g)) * LDB * 4);      \
        GQ_C_RD1(fb1[s], pb, (4 * (g) + 2) * LDB * 4);  \
    } while (0)
    dma_b(0, 0);
    fetch(0, va[0], vb[0]);
    dma_b(1, 1);
    commit(0, va[0], vb[0]);
    fetch(1, va[1], vb[1]);
    fetch(2, va[0], vb[0]);
    if constexpr (BDMA) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // B chunk 0 has landed (behind it: B 1, A 1, A 2)
    __syncthreads();
    unsigned ca0 = aoff0, ca1 = aoff1, cb = boff;  // fragment bases of the current image
    int buf = 0;
    int64_t t = 0;
    GQ_C_READS(0, 0, ca0, ca1, cb);
    GQ_C_READS(1, 1, ca0, ca1, cb);
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // one chunk = groups 0..7; FIRST: its first k-step starts a new chain.  On entry the reads of groups 0 and 1
    // are in flight (sets 0, 1); on exit those of the next chunk are.
    // PAR: parity of the NEXT chunk (t + 1), whose register set is written to LDS and refilled with chunk t + 3
    auto chunk = [&](auto first_c, auto par_c) {
        constexpr bool FIRST = decltype(first_c)::value;
        constexpr int PAR = decltype(par_c)::value;
        const int nbuf = (buf == 2) ? 0 : buf + 1;
        const unsigned na0 = aoff0 + nbuf * STAGE * 4, na1 = aoff1 + nbuf * STAGE * 4, nb = boff + nbuf * STAGE * 4;
#define GQ_C_GROUP(g)                                                                                        \
    do {                                                                                                     \
        if ((g) == 0) dma_b(t + 2, (buf == 0) ? 2 : buf - 1); /* image (t + 2) % 3: everybody left it before barrier t - 1 */ \
        if ((g) == 6) {                                                                                      \
            if constexpr (STAMP) st_a = now();                                                               \
            /* B of chunk t + 1 (issued in chunk t - 1) has landed; behind it: A t+2, B t+2, A t+3 = 6 loads */  \
            if constexpr (BDMA) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                              \
            __builtin_amdgcn_s_barrier(); /* image nbuf is complete: its writes were waited at g = 4 */       \
            if constexpr (STAMP) st_bar += now() - st_a;                                                     \
        }                                                                                                    \
        if ((g) == CG) {                                                                                     \
            if constexpr (STAMP) st_a = now();                                                               \
            GQ_C_WAIT(4, (g) & 3); /* before 6 more LDS operations: lgkmcnt counts to 15 */                  \
            asm volatile("" ::: "memory");                                                                   \
            commit(nbuf, va[PAR], vb[PAR]);                                                                  \
            fetch(t + 3, va[PAR], vb[PAR]);                                                                  \
            asm volatile("" ::: "memory");                                                                   \
            if constexpr (STAMP) st_commit += now() - st_a;                                                  \
        }                                                                                                    \
        if ((g) < 6) GQ_C_READS((g) + 2, ((g) + 2) & 3, ca0, ca1, cb);                                       \
        else GQ_C_READS((g) - 6, ((g) + 2) & 3, na0, na1, nb);                                               \
        GQ_C_WAIT(8, (g) & 3);                                                                               \
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[(g) & 3][0], fb0[(g) & 3], (FIRST && (g) == 0) ? zero : acc[0], 0, 0, 0); \
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[(g) & 3][0], fb0[(g) & 3], (FIRST && (g) == 0) ? zero : acc[1], 0, 0, 0); \
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[(g) & 3][1], fb1[(g) & 3], acc[0], 0, 0, 0);       \
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[(g) & 3][1], fb1[(g) & 3], acc[1], 0, 0, 0);       \
    } while (0)
        GQ_C_GROUP(0); GQ_C_GROUP(1); GQ_C_GROUP(2); GQ_C_GROUP(3);
        GQ_C_GROUP(4); GQ_C_GROUP(5); GQ_C_GROUP(6); GQ_C_GROUP(7);
#undef GQ_C_GROUP
        buf = nbuf;
        ca0 = na0; ca1 = na1; cb = nb;
        ++t;
    };
    auto chain_end = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) cv[i][e] = cv[i][e] - acc[i][e];
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    const int64_t nchain = nk / SPC;
    if constexpr (STAMP) st_t0 = now();
    chunk(T_{}, P1{});  // chunk 0 (even): the next one is odd
    chunk(F_{}, P0{});
    for (int64_t c = 1; c < nchain; ++c) {
        chunk(F_{}, P1{});
        chunk(F_{}, P0{});
        chain_end();
        chunk(T_{}, P1{});
        chunk(F_{}, P0{});
    }
    chunk(F_{}, P1{});
    chunk(F_{}, P0{});
    chain_end();
    if constexpr (STAMP) {
        const unsigned long long t1 = now();
        if (stamps && lane == 0) {
            stamps[4 * wid + 0] = t1 - st_t0; stamps[4 * wid + 1] = st_bar; stamps[4 * wid + 2] = st_commit;
            stamps[4 * wid + 3] = (unsigned long long)nk;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the prefetched (unused) fragments / B pieces of the images after the last
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) cp[i][((e & 3) + 8 * (e >> 2)) * ldc] = cv[i][e];
}
#undef GQ_C_READS

template <int CHAIN, bool BDMA = false>
__global__ __launch_bounds__(512, 2) void gemm32_chain_full_kernel(float* Cmat, int64_t ldc, const float* A, int64_t lda,
                                                                   const float* B, int64_t ldb, int64_t K) {
    g32_chain_full_tile<CHAIN, BDMA>(Cmat, ldc, A, lda, B, ldb, K, (int64_t)blockIdx.y * 128, (int64_t)blockIdx.x * 128);
}
// The same tiles walked by a FIXED number of workgroups (gridDim.x of them, tile t = blockIdx.x + i gridDim.x, column
// tile fastest): the launch never holds more CUs than that, whatever its size -- the look-ahead far update of the
// GPTQ column loop runs next to the loop's own kernels this way (gq_gptq.hip).  Same arithmetic per element.
template <int CHAIN, bool BDMA = false>
__global__ __launch_bounds__(512, 2) void gemm32_chain_full_persistent_kernel(float* Cmat, int64_t ldc, const float* A,
                                                                              int64_t lda, const float* B, int64_t ldb,
                                                                              int64_t K, int64_t ntx, int64_t ntiles) {
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        g32_chain_full_tile<CHAIN, BDMA>(Cmat, ldc, A, lda, B, ldb, K, (t / ntx) * 128, (t % ntx) * 128);
        __syncthreads();  // every wave is done with the LDS ring before the next tile's first image is written
    }
}

// r04, default (option far_bdma): the B chunk by LDS-DMA, A through registers -- the hybrid of the two forms: far alone 0.752 ->
// 0.761 of the fp32 peak, the 4096 x 14336 column loop 11.8 -> 11.5 ms, the far launches inside a step 0.416 -> 0.44 (sum of
// durations), bit-identical (test_trailing_update_kernel_choices_are_bit_identical).
// (Measured and removed, r03/r04: two LDS-DMA forms of the far update -- a four-slot ring with one workgroup per CU,
// 6 % slower than the register-staged chunks above although it issues a tenth of the staging instructions, and a
// two-slot ring with two workgroups per CU, +1.6 % -- DESIGN.md K6; they live in the git history.)
template <int CHAIN>
inline int launch_gemm32_chain_full(float* Cmat, int64_t ldc, const float* A, int64_t lda, const float* B, int64_t ldb,
                                    int64_t M, int64_t N, int64_t K, hipStream_t st, int max_wgs = 0) {
    constexpr int LDS = 3 * G32<128>::STAGE_FLOATS * 4;
    static std::atomic<bool> attr_set{false};  // guards an idempotent call: a race sets the same value twice
    if (!attr_set) {
        GQ_HIP(hipFuncSetAttribute((const void*)gemm32_chain_full_kernel<CHAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        GQ_HIP(hipFuncSetAttribute((const void*)gemm32_chain_full_persistent_kernel<CHAIN>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    const int64_t ntx = N / 128, ntiles = ntx * (M / 128);
    // (measured and removed, r04: EVERY far GEMM as a persistent launch of 256 workgroups -- no workgroup dispatch between tiles --:
    // far alone 0.766-0.770 against 0.760-0.767, the step 91.2-91.8 against 90.0-90.8 ms, the Mixtral block 317 against 314 ms)
    // option far_bdma: the B operand by LDS-DMA (16-byte aligned rows: ldb % 4 == 0 and a 16-byte aligned B)
    const bool bdma = opt(OPT_far_bdma) != 0 && ldb % 4 == 0 && (reinterpret_cast<uintptr_t>(B) % 16 == 0) && 31 * ldb * 4 < (int64_t)1 << 31;
    if (bdma) {
        static std::atomic<bool> attr_b{false};
        if (!attr_b) {
            GQ_HIP(hipFuncSetAttribute((const void*)gemm32_chain_full_kernel<CHAIN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
            GQ_HIP(hipFuncSetAttribute((const void*)gemm32_chain_full_persistent_kernel<CHAIN, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
            attr_b = true;
        }
    }
    if (max_wgs > 0 && ntiles > max_wgs) {
        if (bdma)
            hipLaunchKernelGGL((gemm32_chain_full_persistent_kernel<CHAIN, true>), dim3((unsigned)max_wgs), dim3(512), LDS, st, Cmat,
                               ldc, A, lda, B, ldb, K, ntx, ntiles);
        else
            hipLaunchKernelGGL((gemm32_chain_full_persistent_kernel<CHAIN>), dim3((unsigned)max_wgs), dim3(512), LDS, st, Cmat, ldc,
                               A, lda, B, ldb, K, ntx, ntiles);
    } else {
        dim3 grid((unsigned)ntx, (unsigned)(M / 128)), block(512);
        if (bdma) hipLaunchKernelGGL((gemm32_chain_full_kernel<CHAIN, true>), grid, block, LDS, st, Cmat, ldc, A, lda, B, ldb, K);
        else hipLaunchKernelGGL((gemm32_chain_full_kernel<CHAIN>), grid, block, LDS, st, Cmat, ldc, A, lda, B, ldb, K);
    }
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

// ---- the GPTQ near trailing update: K = 256 (two chains of 128), 64x64 tiles, both operand panels whole in LDS ----
// C = (C - A_0 B_0) - A_1 B_1 with the arithmetic of gemm32_chain_full_kernel<128> (per element: two k-ordered
// 128-long MFMA chains from 0, one subtraction after each).  The near launches of the column loop have M = rows of W,
// N = 256..768 and K = 256: 64..192 tiles of 128x128, each 13.6 us of matrix-pipe time on ONE CU whatever the rest of
// the chip does (measured: ~25 us per launch).  Here a tile is 64x64 (3.4 us) and the launch has 4x the tiles.  What
// made 64-tiles slow in the generic kernel -- a 16 KiB chunk every 0.12 us of MFMA time against ~0.7 us of L2
// latency, one chunk in flight -- is gone: a workgroup asks for its whole A panel [64, 256] and B panel [256, 64] up
// front with 32 global_load_lds_dwordx4 per wave (128 KiB in flight per CU, no VGPR round trip) in four k-quarters,
// and starts the MFMAs of a quarter as soon as that quarter has landed (s_waitcnt vmcnt + barrier per quarter).
//   A in LDS: [quarter][row][16 units of 16 B], unit (row, kq) stored at position kq ^ (row & 15): a lane's
//   ds_read_b128 of (row, 4 k) is conflict-free, the 16 swizzled addresses of a quarter live in 16 VGPRs and the
//   quarter is the instruction's immediate offset.  The lane keeps k = 4 kq + lk and 4 kq + 2 + lk of the four
//   (lk = lane >> 5: the MFMA's k within a step).   B in LDS: [k][64 columns], ds_read_b32 along a row.
// Tiles of one 64-row band of A go to one XCD (blockIdx % 8) so that band crosses the fabric once.
constexpr int NEAR_K = 256, NEAR_A_BYTES = 64 * NEAR_K * 4, NEAR_B_BYTES = NEAR_K * 64 * 4;
constexpr int NEAR_LDS_BYTES = NEAR_A_BYTES + NEAR_B_BYTES;
template <int CHAIN>  // 128 (a template so that every translation unit including this header may instantiate it)
__global__ __launch_bounds__(256, 1) void gemm32_near256_kernel(float* Cmat, int64_t ldc, const float* A, int64_t lda,
                                                                const float* B, int64_t ldb, unsigned nbx, unsigned nby) {
    static_assert(CHAIN == 128, "two chains of 128 k");
    extern __shared__ __attribute__((aligned(16))) float g32_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned id = blockIdx.x;
    unsigned bx, by;
    if (nby % 8 == 0) {
        const unsigned j = id >> 3;
        by = (id & 7) * (nby / 8) + j / nbx;
        bx = j % nbx;
    } else {
        by = id / nbx;
        bx = id % nbx;
    }
    const int64_t m0 = (int64_t)by * 64, n0 = (int64_t)bx * 64;
    const unsigned lds0 = (unsigned)(uintptr_t)g32_smem;
    const int wm = wid >> 1, wn = wid & 1;
    const int li = lane & 31, lk = lane >> 5;
    f32x16 cv;
    float* cp = Cmat + (m0 + wm * 32 + 4 * lk) * ldc + n0 + wn * 32 + li;
#define GQ_N_DL(vo, sp, ldsaddr) \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vo), "s"(sp), "s"(ldsaddr) : "memory")
    {
        // per quarter q (64 k) and wave: 4 instructions of B (k rows 64 q + 16 wid + 4 r .. + 3, whole 256-byte rows) and
        // 4 of A (rows 16 wid + 4 r .. + 3: lane -> row 4 r' + lane / 16, stored position lane % 16 holds unit
        // (lane % 16) ^ (row & 15))
        const float* bp = B + n0;
        const unsigned bvo = (unsigned)(((lane >> 4) * ldb + (lane & 15) * 4) * 4);
        const int arow = lane >> 4;  // row within the instruction's four
        const float* ap = A + m0 * lda;
#define GQ_N_QUARTER(q)                                                                                               \
    do {                                                                                                              \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                               \
            const int j = (q) * 16 + wid * 4 + r; /* 1 KiB piece of the B panel: k rows 4 j .. 4 j + 3 */             \
            const float* sp_ = bp + (int64_t)(4 * j) * ldb;                                                           \
            GQ_N_DL(bvo, sp_, lds0 + NEAR_A_BYTES + (unsigned)j * 1024u);                                             \
        }                                                                                                             \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                               \
            const int row0 = wid * 16 + 4 * r; /* rows row0 .. row0 + 3; row & 15 = 4 r + arow */                     \
            const unsigned avo = (unsigned)((arow * lda + (q) * 64 + (((lane & 15) ^ (4 * r + arow)) << 2)) * 4);     \
            const float* sp_ = ap + (int64_t)row0 * lda;                                                              \
            GQ_N_DL(avo, sp_, lds0 + (unsigned)((q) * 16384 + row0 * 256));                                           \
        }                                                                                                             \
    } while (0)
        GQ_N_QUARTER(0);
#pragma unroll
        for (int e = 0; e < 16; ++e) cv[e] = cp[((e & 3) + 8 * (e >> 2)) * ldc];
        GQ_N_QUARTER(1);
        GQ_N_QUARTER(2);
        GQ_N_QUARTER(3);
#undef GQ_N_QUARTER
    }
#undef GQ_N_DL
    // swizzled LDS addresses of this lane's A row for the 16 units of a quarter
    unsigned aaddr[16];
    {
        const int row = wm * 32 + li;
#pragma unroll
        for (int u = 0; u < 16; ++u) aaddr[u] = lds0 + (unsigned)(row * 256 + ((u ^ (row & 15)) << 4));
    }
    const unsigned bbase = lds0 + NEAR_A_BYTES + (unsigned)(lk * 64 + wn * 32 + li) * 4u;
    g32_u32x4 fa[3];
    float fb0[3], fb1[3];
#define GQ_N_READS(g, s)                                                                                              \
    do {                                                                                                              \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[s]) : "v"(aaddr[(g) & 15]), "n"(((g) >> 4) * 16384) : "memory"); \
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fb0[s]) : "v"(bbase), "n"((g) * 1024) : "memory");         \
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fb1[s]) : "v"(bbase), "n"((g) * 1024 + 512) : "memory");   \
    } while (0)
    // quarter 0 has landed when at most the C loads and quarters 1-3 (16 + 24 operations of this wave) are outstanding
    asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    GQ_N_READS(0, 0);
    GQ_N_READS(1, 1);
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 acc = zero;
    // group g = 4 k = two MFMAs; the reads of group g + 2 are issued first, then the wait leaves exactly those and
    // group g + 1's outstanding (LDS operations return in order).  Two groups before a quarter ends, the next quarter
    // must have landed: its first reads are issued there.
#define GQ_N_GROUP(g)                                                                                            \
    do {                                                                                                         \
        constexpr int s_ = (g) % 3;                                                                              \
        if ((g) % 16 == 14 && (g) < 48) {                                                                        \
            if ((g) == 14) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                                     \
            else if ((g) == 30) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                 \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                \
            __builtin_amdgcn_s_barrier();                                                                        \
        }                                                                                                        \
        if ((g) + 2 < NEAR_K / 4) {                                                                              \
            GQ_N_READS(((g) + 2 < NEAR_K / 4) ? (g) + 2 : 0, ((g) + 2) % 3);                                     \
            asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(fa[s_]), "+v"(fb0[s_]), "+v"(fb1[s_])::"memory");         \
        } else if ((g) + 1 < NEAR_K / 4) {                                                                       \
            asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fa[s_]), "+v"(fb0[s_]), "+v"(fb1[s_])::"memory");         \
        } else {                                                                                                 \
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[s_]), "+v"(fb0[s_]), "+v"(fb1[s_])::"memory");         \
        }                                                                                                        \
        const float a0_ = __builtin_bit_cast(float, lk ? fa[s_].y : fa[s_].x);                                   \
        const float a1_ = __builtin_bit_cast(float, lk ? fa[s_].w : fa[s_].z);                                   \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0_, fb0[s_], ((g) % 32 == 0) ? zero : acc, 0, 0, 0);         \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1_, fb1[s_], acc, 0, 0, 0);                                  \
        if ((g) % 32 == 31) {                                                                                    \
            _Pragma("unroll") for (int e = 0; e < 16; ++e) cv[e] = cv[e] - acc[e];                               \
        }                                                                                                        \
    } while (0)
#define GQ_N_G8(g) GQ_N_GROUP(g); GQ_N_GROUP((g) + 1); GQ_N_GROUP((g) + 2); GQ_N_GROUP((g) + 3); \
                   GQ_N_GROUP((g) + 4); GQ_N_GROUP((g) + 5); GQ_N_GROUP((g) + 6); GQ_N_GROUP((g) + 7)
    GQ_N_G8(0); GQ_N_G8(8); GQ_N_G8(16); GQ_N_G8(24); GQ_N_G8(32); GQ_N_G8(40); GQ_N_G8(48); GQ_N_G8(56);
#undef GQ_N_G8
#undef GQ_N_GROUP
#undef GQ_N_READS
#pragma unroll
    for (int e = 0; e < 16; ++e) cp[((e & 3) + 8 * (e >> 2)) * ldc] = cv[e];
}

// K = 256, M, N multiples of 64, 16-byte aligned operands with ld % 4 == 0
inline int launch_gemm32_near256(float* Cmat, int64_t ldc, const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M,
                                 int64_t N, hipStream_t st) {
    static std::atomic<bool> attr_set{false};
    if (!attr_set) {
        GQ_HIP(hipFuncSetAttribute((const void*)gemm32_near256_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, NEAR_LDS_BYTES));
        attr_set = true;
    }
    const unsigned nbx = (unsigned)(N / 64), nby = (unsigned)(M / 64);
    hipLaunchKernelGGL(gemm32_near256_kernel<128>, dim3(nbx * nby), dim3(256), NEAR_LDS_BYTES, st, Cmat, ldc, A, lda, B, ldb, nbx, nby);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

template <bool TRANS_B, int MODE, bool LOWER, int KR, int CHAIN, int TS, bool FULL>
inline int launch_gemm32_full(float* Cmat, int64_t ldc, const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M,
                            int64_t N, int64_t K, hipStream_t st) {
    constexpr int NW = (CHAIN != 0 && TS == 128) ? 8 : 4;  // the chained kernel also holds the C tile: 8 waves
    static std::atomic<bool> attr_set{false};  // guards an idempotent call: a race sets the same value twice
    if (!attr_set) {
        GQ_HIP(hipFuncSetAttribute((const void*)gemm32_kernel<TRANS_B, MODE, LOWER, KR, CHAIN, TS, FULL, NW>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, G32<TS>::LDS_BYTES));
        attr_set = true;
    }
    dim3 grid((unsigned)((N + TS - 1) / TS), (unsigned)((M + TS - 1) / TS)), block(NW * 64);
    hipLaunchKernelGGL((gemm32_kernel<TRANS_B, MODE, LOWER, KR, CHAIN, TS, FULL, NW>), grid, block, G32<TS>::LDS_BYTES, st, Cmat, ldc,
                       A, lda, B, ldb, M, N, K);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

template <bool TRANS_B, int MODE, bool LOWER, int KR, int CHAIN, int TS>
inline int launch_gemm32_ts(float* Cmat, int64_t ldc, const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M,
                            int64_t N, int64_t K, hipStream_t st) {
    // whole tiles only (every GPTQ / Cholesky shape of a 128-multiple Linear): the unpredicated kernel
    if constexpr (CHAIN == 128 && TS == 128 && !TRANS_B && MODE == 0 && !LOWER && KR == 0) {
        const bool generic = opt(OPT_chain_generic) != 0;  // the generic chained kernel
        if (!generic && M % TS == 0 && N % TS == 0) return launch_gemm32_chain_full<CHAIN>(Cmat, ldc, A, lda, B, ldb, M, N, K, st);
    }
    if (M % TS == 0 && N % TS == 0 && K % TK == 0 && ldc % 4 == 0)
        return launch_gemm32_full<TRANS_B, MODE, LOWER, KR, CHAIN, TS, true>(Cmat, ldc, A, lda, B, ldb, M, N, K, st);
    return launch_gemm32_full<TRANS_B, MODE, LOWER, KR, CHAIN, TS, false>(Cmat, ldc, A, lda, B, ldb, M, N, K, st);
}

template <bool TRANS_B, int MODE, bool LOWER, int KR = 0, int CHAIN = 0>
inline int launch_gemm32(float* Cmat, int64_t ldc, const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M,
                         int64_t N, int64_t K, hipStream_t st) {
    if (M <= 0 || N <= 0 || K <= 0) return GQ_OK;
    if ((lda % 4) || (ldb % 4) || ((uintptr_t)A % 16) || ((uintptr_t)B % 16))
        GQ_FAIL(GQ_E_BAD_SHAPE, "gemm32: A/B must be 16-byte aligned with ld %% 4 == 0");
    if (CHAIN != 0 && (K % CHAIN)) GQ_FAIL(GQ_E_BAD_SHAPE, "gemm32: K=%ld is not a multiple of the chain length %d", (long)K, CHAIN);
    // fewer 128-tiles than CUs: 64-tiles put four times as many CUs on the (latency-bound) problem.  Every output
    // element is the same k-ordered chain either way, so the choice never changes a result.
    const int64_t tiles128 = ((M + 127) / 128) * ((N + 127) / 128) / (LOWER ? 2 : 1);
    const int64_t max64 = opt(OPT_gemm32_64_max);  // 0: never
    if (CHAIN == 0 && tiles128 < max64)
        return launch_gemm32_ts<TRANS_B, MODE, LOWER, KR, CHAIN == 0 ? 0 : 0, 64>(Cmat, ldc, A, lda, B, ldb, M, N, K, st);
    return launch_gemm32_ts<TRANS_B, MODE, LOWER, KR, CHAIN, 128>(Cmat, ldc, A, lda, B, ldb, M, N, K, st);
}

}  // namespace gq
