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
// 16-bit path: X[T,C] is first re-laid out (transpose16_kernel) into per-(panel, stage) blocks
// that ARE the LDS image of an operand stage, so both MFMA operands are 16-byte K-contiguous
// fragments and the operand stream is perfectly sequential in HBM.
#include "gq_common.hpp"

namespace gq {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------- transpose
// X[T,C] (16-bit) -> Xb[C/128 panels][Tp/64 stages][128 rows][64 k]: each (panel, stage) block is
// the 16 KiB LDS image of one SYRK operand stage, contiguous in HBM and ALREADY XOR-swizzled
// (16-byte chunk kc of row r is stored at chunk kc ^ ((r >> 1) & 7)), so the SYRK kernel streams it
// with lane-linear global_load_lds: every wave instruction copies 1 KiB of consecutive bytes.
// Tokens t >= T are zero padding.  One workgroup = one block; 16-byte global accesses on both sides.
__global__ __launch_bounds__(256) void transpose16_kernel(const uint16_t* __restrict__ X, int64_t T, int64_t C,
                                                          uint16_t* __restrict__ Xb, int64_t nstage) {
    __shared__ __attribute__((aligned(16))) uint16_t tile[64][128 + 8];  // [t][c], row = 272 B
    const int64_t st = blockIdx.x, pn = blockIdx.y;
    const int64_t t0 = st * 64, c0 = pn * 128;
    const int tid = threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int idx = tid + u * 256, tr = idx >> 4, c8 = (idx & 15) * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (t0 + tr < T) v = *reinterpret_cast<const uint4*>(X + (t0 + tr) * C + c0 + c8);
        *reinterpret_cast<uint4*>(&tile[tr][c8]) = v;
    }
    __syncthreads();
    uint16_t* out = Xb + (pn * nstage + st) * (128 * 64);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int p = tid + u * 256, r = p >> 3, kc = (p & 7) ^ ((r >> 1) & 7);  // stored chunk p holds k-chunk kc
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            w[e] = (uint32_t)tile[kc * 8 + 2 * e][r] | ((uint32_t)tile[kc * 8 + 2 * e + 1][r] << 16);
        *reinterpret_cast<uint4*>(out + p * 8) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// ------------------------------------------------------------ 16-bit SYRK
// Grouped launch: up to 8 problems (the distinct Linear inputs of one transformer
// block) share one grid, so the tile count is >> 256 CUs even when C = 4096 gives only
// 528 upper-triangular tiles per Hessian.
//
// Workgroup = 256 threads = 4 waves (2x2), tile 128x128, wave tile 64x64 = 2x2
// v_mfma_f32_32x32x16_{f16,bf16}.  K streams in stages of 64 elements (128 B per row)
// through a double-buffered 64 KiB LDS image filled with global_load_lds_dwordx4
// (16 B per lane, no VGPR round trip).  The LDS image is row-major [128][128 B] with the
// eight 16-B chunks of a row XOR-swizzled by ((row >> 1) & 7) -- row parity already picks the
// half of the 256-B bank row, so the 16 rows of a ds_read_b128 lane group land on 16 distinct
// 16-B slots (conflict-free; (row & 7) measured 2-way).  global_load_lds writes lane-linear,
// so the swizzle is applied to the per-lane SOURCE address (8 consecutive lanes still
// fetch one full 128-B line) and again on the ds_read_b128 fragment address.
constexpr int HT = 128;  // output tile
constexpr int HK = 64;   // k per stage (16-bit elements) = 128 B per row
constexpr int H_STAGE_BYTES = 2 * HT * HK * 2;  // A + B = 32 KiB
constexpr int H_MAX_GROUP = 8;

struct SyrkProblem {
    float* H;
    const uint16_t* Xt;
    int64_t C, Tp;
    float beta, alpha;
    int tile_begin, nt;
};
struct SyrkGroup {
    int n, total_tiles;
    SyrkProblem p[H_MAX_GROUP];
};

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

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <bool BF16>
__global__ __launch_bounds__(256, 2) void syrk16_kernel(const SyrkGroup grp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 stages x (A 16 KiB | B 16 KiB)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    // Tile order.  A 128x128 tile streams two 128-row panels of Xt; at 2 workgroups per CU the
    // 64 workgroups resident on one XCD (block b runs on XCD b % 8) are mapped to ONE 8x8
    // super-tile of the upper triangle, so together they stream 16 panels instead of 65 and the
    // XCD's L2 serves the other 3/4 of the operand traffic (HBM-bound -> MFMA/L2-bound).
    const int xcd = blockIdx.x & 7, kx = blockIdx.x >> 3;
    const int g = (kx >> 6) * 8 + xcd, slot = kx & 63;
    if (g >= grp.total_tiles) return;  // total_tiles counts SUPER-tiles here
    int pi = 0;
    for (int i = 1; i < grp.n; ++i)
        if (g >= grp.p[i].tile_begin) pi = i;
    const SyrkProblem& P = grp.p[pi];
    int64_t sI, sJ;
    tri_tile(g - P.tile_begin, (P.nt + 7) >> 3, sI, sJ);
    const int64_t ti = sI * 8 + (slot >> 3), tj = sJ * 8 + (slot & 7);
    if (ti >= P.nt || tj >= P.nt || ti > tj) return;
    const int64_t i0 = ti * HT, j0 = tj * HT, Tp = P.Tp, C = P.C;
    const uint16_t* __restrict__ Xt = P.Xt;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // operand stage (panel, s) is the contiguous, pre-swizzled 16 KiB block Xb[(panel*nstage + s)*8192 ...]
    const int64_t nk = Tp / HK;
    const uint16_t* srcA = Xt + (ti * nk) * (HT * HK) + tid * 8;
    const uint16_t* srcB = Xt + (tj * nk) * (HT * HK) + tid * 8;
    auto stage = [&](int buf, int64_t s) {
        unsigned char* base = smem + buf * H_STAGE_BYTES;
        const uint16_t* pa = srcA + s * (HT * HK);
        const uint16_t* pb = srcB + s * (HT * HK);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            // LDS destination = wave-uniform base + lane*16 (hardware adds the lane offset)
            unsigned char* la = base + (t * 256 + wid * 64) * 16;
            __builtin_amdgcn_global_load_lds((glb_void*)(pa + t * 2048), (lds_void*)la, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void*)(pb + t * 2048), (lds_void*)(la + HT * HK * 2), 16, 0, 0);
        }
    };
    const int li = lane & 31, lk = lane >> 5;
    int offA[2], offB[2], swz[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra = wm * 64 + i * 32 + li, rb = wn * 64 + i * 32 + li;
        offA[i] = ra * 128;
        offB[i] = HT * HK * 2 + rb * 128;
        swz[0][i] = (ra >> 1) & 7;
        swz[1][i] = (rb >> 1) & 7;
    }
    stage(0, 0);
    __syncthreads();
    for (int64_t t = 0; t < nk; ++t) {
        // 1) pull ALL fragments of stage t into registers while no LDS-DMA is in flight (hipcc
        //    conservatively drains vmcnt before any ds_read that follows a global_load_lds)
        const unsigned char* base = smem + (t & 1) * H_STAGE_BYTES;
        uint4 a[4][2], b[4][2];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int kc = s4 * 2 + lk;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[s4][i] = *reinterpret_cast<const uint4*>(base + offA[i] + ((kc ^ swz[0][i]) << 4));
                b[s4][i] = *reinterpret_cast<const uint4*>(base + offB[i] + ((kc ^ swz[1][i]) << 4));
            }
        }
        // 2) start the DMA of stage t+1 into the other buffer (last read one barrier ago)
        if (t + 1 < nk) stage((int)((t + 1) & 1), t + 1);
        __builtin_amdgcn_sched_barrier(0);  // keep the DMA issue ahead of the MFMA block ...
        // 3) 16 MFMAs per wave cover the DMA flight
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma16<BF16>(a[s4][i], b[s4][j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);  // ... and the MFMA block ahead of the vmcnt(0) drain
        __syncthreads();  // vmcnt(0) + barrier: stage t+1 landed, stage t's buffer is free
    }
    float* __restrict__ H = P.H;
    const float beta = P.beta, alpha = P.alpha;
    const int lc = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t col = j0 + wn * 64 + j * 32 + lc;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t row = i0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                float h = beta * H[row * C + col] + alpha * acc[i][j][e];
                H[row * C + col] = h;
                if (ti != tj) H[col * C + row] = h;  // mirror (H stays exactly symmetric)
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

int h_accumulate_grouped(int n, float* const* H, const void* const* X, const int64_t* T, const int64_t* C,
                         const float* beta, const float* alpha, int x_dtype, void* ws, size_t ws_bytes,
                         hipStream_t st) {
    if (n <= 0 || n > H_MAX_GROUP) GQ_FAIL(GQ_E_BAD_SHAPE, "gq_h_accumulate_grouped: n=%d not in 1..%d", n, H_MAX_GROUP);
    if (!H || !X || !T || !C || !beta || !alpha) GQ_FAIL(GQ_E_NULL, "gq_h_accumulate_grouped: null pointer");
    const dim3 block(256);
    if (x_dtype == GQ_F32) {
        for (int i = 0; i < n; ++i) {
            if (T[i] <= 0 || C[i] <= 0 || (C[i] % HT)) GQ_FAIL(GQ_E_BAD_SHAPE, "gq_h_accumulate: T=%ld C=%ld (C %% 128 != 0)", (long)T[i], (long)C[i]);
            const int64_t nt = C[i] / HT;
            ProfScope ps(PT_SYRK, st);
            hipLaunchKernelGGL(syrk32_kernel, dim3((unsigned)(nt * (nt + 1) / 2)), block, 0, st, H[i],
                               C[i], (const float*)X[i], T[i], beta[i], alpha[i]);
            GQ_LAUNCH_CHECK();
        }
        return GQ_OK;
    }
    if (x_dtype != GQ_F16 && x_dtype != GQ_BF16) GQ_FAIL(GQ_E_BAD_TYPE, "gq_h_accumulate: unknown x_dtype %d", x_dtype);
    size_t need = 0;
    for (int i = 0; i < n; ++i) {
        if (!H[i] || !X[i]) GQ_FAIL(GQ_E_NULL, "gq_h_accumulate: null pointer");
        if (T[i] <= 0 || C[i] <= 0 || (C[i] % HT)) GQ_FAIL(GQ_E_BAD_SHAPE, "gq_h_accumulate: T=%ld C=%ld (C %% 128 != 0)", (long)T[i], (long)C[i]);
        need += h_accumulate_workspace_bytes(T[i], C[i]);
    }
    if (!ws || ws_bytes < need) GQ_FAIL(GQ_E_WORKSPACE, "gq_h_accumulate: workspace %zu < %zu bytes", ws_bytes, need);
    SyrkGroup grp;
    grp.n = n;
    int tiles = 0;
    unsigned char* wp = reinterpret_cast<unsigned char*>(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    for (int i = 0; i < n; ++i) {
        const int64_t Tp = (T[i] + HK - 1) / HK * HK;
        uint16_t* Xt = reinterpret_cast<uint16_t*>(wp);
        wp += ((size_t)C[i] * Tp * 2 + 255) & ~(size_t)255;
        {
            ProfScope ps(PT_TRANSPOSE, st);
            dim3 tg((unsigned)(Tp / HK), (unsigned)(C[i] / HT));
            hipLaunchKernelGGL(transpose16_kernel, tg, block, 0, st, (const uint16_t*)X[i], T[i], C[i], Xt, Tp / HK);
            GQ_LAUNCH_CHECK();
        }
        const int nt = (int)(C[i] / HT);
        grp.p[i] = SyrkProblem{H[i], Xt, C[i], Tp, beta[i], alpha[i], tiles, nt};
        const int ns = (nt + 7) / 8;  // 8x8 super-tiles per dimension
        tiles += ns * (ns + 1) / 2;
    }
    grp.total_tiles = tiles;
    static bool attr_set = false;
    if (!attr_set) {
        GQ_HIP(hipFuncSetAttribute((const void*)syrk16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * H_STAGE_BYTES));
        GQ_HIP(hipFuncSetAttribute((const void*)syrk16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * H_STAGE_BYTES));
        attr_set = true;
    }
    ProfScope ps(PT_SYRK, st);
    if (x_dtype == GQ_BF16)
        hipLaunchKernelGGL(syrk16_kernel<true>, dim3((unsigned)((tiles + 7) / 8 * 8 * 64)), block, 2 * H_STAGE_BYTES, st, grp);
    else
        hipLaunchKernelGGL(syrk16_kernel<false>, dim3((unsigned)((tiles + 7) / 8 * 8 * 64)), block, 2 * H_STAGE_BYTES, st, grp);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

int h_accumulate(float* H, const void* X, int x_dtype, int64_t T, int64_t C, float beta, float alpha, void* ws,
                 size_t ws_bytes, hipStream_t st) {
    return h_accumulate_grouped(1, &H, &X, &T, &C, &beta, &alpha, x_dtype, ws, ws_bytes, st);
}

}  // namespace gq
