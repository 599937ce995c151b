// gq_gemm3p.hpp -- fp32-accurate GEMM on the 16-bit matrix cores from PRE-SPLIT operand images, for the large
// GEMMs of the blocked Cholesky / triangular inverse (K3).  Successor of gq_gemm3b.hpp for the top recursion levels.
//
// gq_gemm3b.hpp splits every fp32 operand element into three bf16 terms on the VALU each time a tile loads it and
// moves it global -> VGPR -> LDS as fp32 (PMC, r02: 52 % MFMA busy).  Here every operand is split ONCE, by a
// bandwidth-bound pass, into an IMAGE: per (128-row panel, 32-k chunk) NP planes of [128 rows][32 k] 16-bit values,
// 8 KiB each, already in the swizzled layout of an LDS ring slot.  The GEMM is then the SYRK's machine
// (gq_hessian.hip, syrk16_256e_kernel): 256x256 tiles, 8 waves with 128x64 wave tiles of v_mfma_f32_16x16x32,
// a ring of four 32 KiB LDS slots filled by global_load_lds_dwordx4 (no VGPR round trip, no ds_write, no VALU in
// the loop), hand-ordered fragment reads / MFMAs / waits.  A "half-stage" is one (A plane, B plane) pair of one
// k32 chunk; the product a*b is the sum of NPROD half-stages per chunk:
//
//   NP = 2, fp16 (default): every operand ROW is scaled by a power of two so that its largest entry lies in
//     [2^14, 2^15), then x = x1 + x2 + r with x1 = fp16(x), x2 = fp16(x - x1), |r| <= 2^-22 |x| (or 2^-40 of the row
//     maximum where x2 is denormal); products a1b2, a2b1, a1b1 (each exact in fp32), a2b2 <= 2^-22 |ab| dropped; the
//     epilogue undoes the scales.  THREE MFMA products per term.  The error bound is normwise per row, which is why
//     gq_h_prepare equilibrates the matrix first (gq_cholesky.hip, equil_kernel: all k of a product weigh the same);
//     measured MORE accurate than the exact bf16 split below (two roundings to nearest beat three truncations).
//   NP = 3, bf16 (GQ_CHOL_BF16X3=1): x = x1 + x2 + x3 EXACTLY (truncation split, 8 + 8 + 8 significand bits); the six
//     products of weight >= 2^-16 -- a1b3, a3b1, a2b2, a1b2, a2b1, a1b1, smallest first, each exact in fp32 -- are
//     accumulated in fp32: the dropped terms are <= 2^-22 |ab|, the size of an fp32 rounding (as gq_gemm3b.hpp).
//
// Triangular operands: k-ranges are skipped per tile (never multiplied), ranges always start at image chunk 0 --
// operands whose valid range ENDS at K are imaged with their chunks in reverse order (both operands of the product
// alike), so all tiles of a launch stream the same panels at the same time.
// Scheduling (make_plan): work units = (tile, chunk range), assigned STATICALLY: the tiles in super-tile order (8 x 4
// tiles share 12 operand panels) form one sequence, cut into 8 equal-work segments (one per XCD); inside an XCD whole
// rounds of 32, then the wrap-around rule levels the workgroups, cutting a tile along k where it overflows.  Cut tiles
// store raw fp32 sums into slots of a partial buffer and reduce_kernel adds them in fixed order (deterministic).
// Tolerance-class like gq_gemm3b.hpp (U = chol(H^-1) is checked against fp64); the GPTQ trailing update -- the
// bit-exact parity gate -- never comes here.
#pragma once
#include <algorithm>
#include <atomic>
#include <vector>

#include "gq_common.hpp"

namespace gq {
namespace p3 {

constexpr int BLK = 8192;                 // one image block: [128 rows][32 k] 16-bit
constexpr int TILE = 256;
constexpr int SLOT_BYTES = 4 * BLK;       // A: 2 panels | B: 2 panels
constexpr int LDS_BYTES = 4 * SLOT_BYTES; // ring of four: 128 KiB
constexpr int CPT = 8;                    // chunks per 256 k
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ __forceinline__ unsigned swz(int r) { return (0x78u >> (2 * ((r >> 2) & 3))) & 3u; }

// ------------------------------------------------------------------------------------------------ split passes
// Op[r][k], r < 128 * gridDim.y, k < 32 * KC:  TRANS ? S[k * ld + r] : S[r * ld + k]  ->  image
// img[((panel * KC + (rev ? KC-1-c : c)) * NP + plane) * BLK + (r * 4 + (kc ^ swz(r))) * 16 + ...]
// tri: 1 = Op is zero where (k >> 7) > (r >> 7) (those source blocks are never read: they may hold garbage),
//      2 = zero where (k >> 7) < (r >> 7).
// NP == 2: rmax[r] = bits of max_k |Op[r][k]| (row_absmax_kernel); inv_scale[r] = 2^-e is written for the epilogue.
__device__ __forceinline__ void split3(const float (&x)[8], uint4& p1, uint4& p2, uint4& p3) {
    unsigned a[8], b[8], c[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const unsigned xb = __builtin_bit_cast(unsigned, x[e]);
        const unsigned x1 = xb & 0xffff0000u;
        const float xr = x[e] - __builtin_bit_cast(float, x1);
        const unsigned x2 = __builtin_bit_cast(unsigned, xr) & 0xffff0000u;
        const float xs = xr - __builtin_bit_cast(float, x2);  // <= 8 significant bits: exact in bf16
        a[e] = x1; b[e] = x2; c[e] = __builtin_bit_cast(unsigned, xs);
    }
    // v_perm_b32: upper halves of (hi, lo)
    p1 = make_uint4(__builtin_amdgcn_perm(a[1], a[0], 0x07060302u), __builtin_amdgcn_perm(a[3], a[2], 0x07060302u),
                    __builtin_amdgcn_perm(a[5], a[4], 0x07060302u), __builtin_amdgcn_perm(a[7], a[6], 0x07060302u));
    p2 = make_uint4(__builtin_amdgcn_perm(b[1], b[0], 0x07060302u), __builtin_amdgcn_perm(b[3], b[2], 0x07060302u),
                    __builtin_amdgcn_perm(b[5], b[4], 0x07060302u), __builtin_amdgcn_perm(b[7], b[6], 0x07060302u));
    p3 = make_uint4(__builtin_amdgcn_perm(c[1], c[0], 0x07060302u), __builtin_amdgcn_perm(c[3], c[2], 0x07060302u),
                    __builtin_amdgcn_perm(c[5], c[4], 0x07060302u), __builtin_amdgcn_perm(c[7], c[6], 0x07060302u));
}
__device__ __forceinline__ float scale_from_max(unsigned maxbits, float& inv) {
    // largest entry -> [2^14, 2^15): exponent e = 14 - floor(log2 max); zero / denormal rows keep their values
    const int ex = (int)((maxbits >> 23) & 0xff);
    int e = (ex == 0 || ex == 0xff) ? 0 : 14 - (ex - 127);
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
    inv = __builtin_bit_cast(float, (unsigned)(127 - e) << 23);
    return __builtin_bit_cast(float, (unsigned)(127 + e) << 23);
}
__device__ __forceinline__ void split2h(const float (&x)[8], float mult, uint4& p1, uint4& p2) {
    unsigned a[4], b[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned lo[2], hi[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float v = x[2 * e + u] * mult;  // exact (power of two), |v| < 2^15
            asm volatile("" : "+v"(v));     // keep the fp32 value: no fused convert of the product
            const _Float16 h1 = (_Float16)v;
            float r = v - (float)h1;        // exact
            asm volatile("" : "+v"(r));
            const _Float16 h2 = (_Float16)r;
            lo[u] = (unsigned)__builtin_bit_cast(uint16_t, h1);
            hi[u] = (unsigned)__builtin_bit_cast(uint16_t, h2);
        }
        a[e] = lo[0] | (lo[1] << 16);
        b[e] = hi[0] | (hi[1] << 16);
    }
    p1 = make_uint4(a[0], a[1], a[2], a[3]);
    p2 = make_uint4(b[0], b[1], b[2], b[3]);
}

template <int NP, bool TRANS>
__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ S, int64_t ld, int KC, int tri, int rev,
                                                    unsigned char* __restrict__ img, const unsigned* __restrict__ rmax,
                                                    float* __restrict__ inv_scale) {
    __shared__ float T[TRANS ? 32 : 1][TRANS ? 132 : 1];
    const int c = blockIdx.x, pn = blockIdx.y, tid = threadIdx.x;
    const int kblk = c >> 2;
    const bool zero = (tri == 1 && kblk > pn) || (tri == 2 && kblk < pn);
    if constexpr (TRANS) {
        if (!zero) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = tid + u * 256, kk = idx >> 5, r4 = (idx & 31) * 4;
                const float4 v = *reinterpret_cast<const float4*>(S + (int64_t)(32 * c + kk) * ld + 128 * pn + r4);
                T[kk][r4] = v.x; T[kk][r4 + 1] = v.y; T[kk][r4 + 2] = v.z; T[kk][r4 + 3] = v.w;
            }
        }
        __syncthreads();
    }
    unsigned char* out = img + ((size_t)pn * KC + (rev ? KC - 1 - c : c)) * (size_t)(NP * BLK);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int p = tid + u * 256, r = p >> 2, kc = (p & 3) ^ (int)swz(r);
        float x[8];
        if (zero) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = 0.0f;
        } else if constexpr (TRANS) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = T[8 * kc + e][r];
        } else {
            const float* src = S + (int64_t)(128 * pn + r) * ld + 32 * c + 8 * kc;
            const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
            x[0] = v0.x; x[1] = v0.y; x[2] = v0.z; x[3] = v0.w; x[4] = v1.x; x[5] = v1.y; x[6] = v1.z; x[7] = v1.w;
        }
        if constexpr (NP == 3) {
            uint4 p1, p2, p3;
            split3(x, p1, p2, p3);
            *reinterpret_cast<uint4*>(out + p * 16) = p1;
            *reinterpret_cast<uint4*>(out + BLK + p * 16) = p2;
            *reinterpret_cast<uint4*>(out + 2 * BLK + p * 16) = p3;
        } else {
            float inv;
            const float mult = scale_from_max(rmax[128 * pn + r], inv);
            if (c == 0 && (p & 3) == 0) inv_scale[128 * pn + r] = inv;
            uint4 p1, p2;
            split2h(x, mult, p1, p2);
            *reinterpret_cast<uint4*>(out + p * 16) = p1;
            *reinterpret_cast<uint4*>(out + BLK + p * 16) = p2;
        }
    }
}

// rmax[r] = max(rmax[r], bits of max_k |Op[r][k]|) over this workgroup's k-slab of 128 (rmax zeroed by the caller)
template <bool TRANS>
__global__ __launch_bounds__(256) void row_absmax_kernel(const float* __restrict__ S, int64_t ld, int tri,
                                                         unsigned* __restrict__ rmax) {
    const int kb = blockIdx.x, pn = blockIdx.y, tid = threadIdx.x;  // 128 x 128 block (kb = k / 128)
    if ((tri == 1 && kb > pn) || (tri == 2 && kb < pn)) return;
    __shared__ unsigned part[128];
    if (tid < 128) part[tid] = 0u;
    __syncthreads();
    if constexpr (TRANS) {
        // thread = (k row kk = tid >> 5 (+8 u), 4 consecutive r)
        float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
        const int r4 = (tid & 31) * 4;
#pragma unroll 4
        for (int kk = tid >> 5; kk < 128; kk += 8) {
            const float4 v = *reinterpret_cast<const float4*>(S + (int64_t)(128 * kb + kk) * ld + 128 * pn + r4);
            m0 = fmaxf(m0, fabsf(v.x)); m1 = fmaxf(m1, fabsf(v.y)); m2 = fmaxf(m2, fabsf(v.z)); m3 = fmaxf(m3, fabsf(v.w));
        }
        atomicMax(&part[r4], __builtin_bit_cast(unsigned, m0));
        atomicMax(&part[r4 + 1], __builtin_bit_cast(unsigned, m1));
        atomicMax(&part[r4 + 2], __builtin_bit_cast(unsigned, m2));
        atomicMax(&part[r4 + 3], __builtin_bit_cast(unsigned, m3));
    } else {
        // thread = (row r = tid >> 1 ... 128 rows, half h = tid & 1: 64 floats)
        const int r = tid >> 1, h = tid & 1;
        const float* src = S + (int64_t)(128 * pn + r) * ld + 128 * kb + 64 * h;
        float m = 0.f;
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(src + 4 * q);
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
        atomicMax(&part[r], __builtin_bit_cast(unsigned, m));
    }
    __syncthreads();
    if (tid < 128 && part[tid]) atomicMax(&rmax[128 * pn + tid], part[tid]);
}

// ------------------------------------------------------------------------------------------------ the GEMM
struct Problem {
    const unsigned char* Aimg;
    const unsigned char* Bimg;
    uint32_t a_panel_bytes, b_panel_bytes;  // KC_image * NP * BLK
    float* C;
    int64_t ldc;
    int mode;  // 0: C -= P, 1: C = P, 2: C = -P
    const float* row_scale;  // NP == 2 only (nullptr: none)
    const float* col_scale;
    float* partial;  // K-split slots [slot][256][256]
};
struct Group {
    Problem p[2];
    const uint32_t* table;  // [0..257): unit list bounds of the 256 workgroups (workgroup b = XCD b % 8), then the units
                            // (4 words each: prob << 28 | tm << 14 | tn, chunk begin, chunk end, slot + 1 or 0)
};
constexpr int T_UNITS = 260, NWG = 256;

template <int NP>
__global__ __launch_bounds__(512, 2) void gemm_kernel(const Group g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const unsigned qb = g.table[blockIdx.x], qe = g.table[blockIdx.x + 1];
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const unsigned ldsw = lds0 + (unsigned)wid * 1024u;  // + slot * 32 KiB + piece * 8 KiB; hardware adds lane * 16
    const int lr = lane & 15, lk = lane >> 4;
    unsigned bA0, bA1, bB0, bB1;
    {
        const unsigned ch = ((unsigned)lk ^ swz(lr)) << 4;
        bA0 = lds0 + (unsigned)(wm * 128 + lr) * 64u + ch;
        bB0 = lds0 + 16384u + (unsigned)(wn * 64 + lr) * 64u + ch;
        bA1 = bA0 + 65536u;
        bB1 = bB0 + 65536u;
    }
    for (unsigned ui = qb; ui < qe; ++ui) {
        __syncthreads();  // every wave is done with the ring of the previous unit
        const uint32_t* uw = g.table + T_UNITS + 4 * (size_t)ui;
        const uint32_t w0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)uw[0]);
        const int cb = __builtin_amdgcn_readfirstlane((int)uw[1]), ce = __builtin_amdgcn_readfirstlane((int)uw[2]);
        const uint32_t w3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)uw[3]);
        const Problem& P = g.p[w0 >> 28];
        const int tm = (int)((w0 >> 14) & 0x3fff), tn = (int)(w0 & 0x3fff);
        // wave-uniform plane streams (SGPR pairs) + per-lane byte offsets of the chunks in flight
        const char* a0 = reinterpret_cast<const char*>(P.Aimg) + (size_t)(2 * tm) * P.a_panel_bytes;
        const char* a1 = a0 + P.a_panel_bytes;
        const char* b0 = reinterpret_cast<const char*>(P.Bimg) + (size_t)(2 * tn) * P.b_panel_bytes;
        const char* b1 = b0 + P.b_panel_bytes;
        const char *a0p0 = a0, *a0p1 = a0 + BLK, *a1p0 = a1, *a1p1 = a1 + BLK;
        const char *b0p0 = b0, *b0p1 = b0 + BLK, *b1p0 = b1, *b1p1 = b1 + BLK;
        const char *a0p2 = a0 + 2 * BLK, *a1p2 = a1 + 2 * BLK, *b0p2 = b0 + 2 * BLK, *b1p2 = b1 + 2 * BLK;
        (void)a0p2; (void)a1p2; (void)b0p2; (void)b1p2;
        constexpr unsigned CH = NP * BLK;
        const unsigned vlast = (unsigned)(ce - 1) * CH + (unsigned)tid * 16u;
        unsigned V0 = (unsigned)cb * CH + (unsigned)tid * 16u, V1, V2, V3, V4;
#define P3_VNEXT()                                               \
    V1 = V0 + CH < vlast ? V0 + CH : vlast;                      \
    V2 = V0 + 2 * CH < vlast ? V0 + 2 * CH : vlast;              \
    V3 = V0 + 3 * CH < vlast ? V0 + 3 * CH : vlast;              \
    V4 = V0 + 4 * CH < vlast ? V0 + 4 * CH : vlast
        P3_VNEXT();
        (void)V3; (void)V4;
#define P3_DL(vo, sp, slot, part)                                                     \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"      \
                 :: "v"(vo), "s"(sp), "s"(ldsw + (unsigned)((slot) * SLOT_BYTES + (part) * BLK)) : "memory")
#define P3_DL4(vo, PA, PB, slot)                                                      \
    P3_DL(vo, a0p##PA, slot, 0); P3_DL(vo, a1p##PA, slot, 1); P3_DL(vo, b0p##PB, slot, 2); P3_DL(vo, b1p##PB, slot, 3)
        f32x4 c00, c01, c02, c03, c10, c11, c12, c13, c20, c21, c22, c23, c30, c31, c32, c33, c40, c41, c42, c43, c50,
            c51, c52, c53, c60, c61, c62, c63, c70, c71, c72, c73;
#define P3_Z(c) c = f32x4{0.f, 0.f, 0.f, 0.f}
        P3_Z(c00); P3_Z(c01); P3_Z(c02); P3_Z(c03); P3_Z(c10); P3_Z(c11); P3_Z(c12); P3_Z(c13);
        P3_Z(c20); P3_Z(c21); P3_Z(c22); P3_Z(c23); P3_Z(c30); P3_Z(c31); P3_Z(c32); P3_Z(c33);
        P3_Z(c40); P3_Z(c41); P3_Z(c42); P3_Z(c43); P3_Z(c50); P3_Z(c51); P3_Z(c52); P3_Z(c53);
        P3_Z(c60); P3_Z(c61); P3_Z(c62); P3_Z(c63); P3_Z(c70); P3_Z(c71); P3_Z(c72); P3_Z(c73);
#undef P3_Z
        u32x4 pa0, pa1, pa2, pa3, pa4, pa5, pa6, pa7, pb0, pb1, pb2, pb3;
        u32x4 qa0, qa1, qa2, qa3, qa4, qa5, qa6, qa7, qb0, qb1, qb2, qb3;
#define P3_DSR(dst, base, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(off) : "memory")
#define P3_LGKM1(N, x) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(x)::"memory")
#define P3_LGKM2(N, x, y) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(x), "+v"(y)::"memory")
#define P3_MF(c, a, b)                                                                                          \
    do {                                                                                                        \
        if constexpr (NP == 3) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); \
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));                 \
    } while (0)
        // one half-stage: 32 MFMAs on register set X; the 12 fragment reads of the next half-stage into set Y behind
        // MFMAs 1, 3, ..., 23 (reads are issued a0,b0,b1,b2,b3,a1..a7; lgkmcnt(k) = all but the youngest k are back);
        // the barrier behind MFMA 0; the four DMA pieces of the half-stage three ahead behind MFMAs 13/17/21/25
#define P3_STEP(X, Y, AY, BY, OFF, MID, L0, L1, L2, L3)                                               \
    do {                                                                                              \
        P3_LGKM2(10, X##a0, X##b0);                                                                   \
        P3_MF(c00, X##a0, X##b0);                                                                     \
        MID;                                                                                          \
        P3_LGKM1(9, X##b1);                                                                           \
        P3_MF(c01, X##a0, X##b1);                                                                     \
        P3_DSR(Y##a0, AY, (OFF) + 0);                                                                 \
        P3_LGKM1(9, X##b2);                                                                           \
        P3_MF(c02, X##a0, X##b2);                                                                     \
        P3_LGKM1(8, X##b3);                                                                           \
        P3_MF(c03, X##a0, X##b3);                                                                     \
        P3_DSR(Y##b0, BY, (OFF) + 0);                                                                 \
        P3_LGKM1(8, X##a1);                                                                           \
        P3_MF(c10, X##a1, X##b0);                                                                     \
        P3_MF(c11, X##a1, X##b1);                                                                     \
        P3_DSR(Y##b1, BY, (OFF) + 1024);                                                              \
        P3_MF(c12, X##a1, X##b2);                                                                     \
        P3_MF(c13, X##a1, X##b3);                                                                     \
        P3_DSR(Y##b2, BY, (OFF) + 2048);                                                              \
        P3_LGKM1(9, X##a2);                                                                           \
        P3_MF(c20, X##a2, X##b0);                                                                     \
        P3_MF(c21, X##a2, X##b1);                                                                     \
        P3_DSR(Y##b3, BY, (OFF) + 3072);                                                              \
        P3_MF(c22, X##a2, X##b2);                                                                     \
        P3_MF(c23, X##a2, X##b3);                                                                     \
        P3_DSR(Y##a1, AY, (OFF) + 1024);                                                              \
        P3_LGKM1(10, X##a3);                                                                          \
        P3_MF(c30, X##a3, X##b0);                                                                     \
        P3_MF(c31, X##a3, X##b1);                                                                     \
        P3_DSR(Y##a2, AY, (OFF) + 2048);                                                              \
        L0;                                                                                           \
        P3_MF(c32, X##a3, X##b2);                                                                     \
        P3_MF(c33, X##a3, X##b3);                                                                     \
        P3_DSR(Y##a3, AY, (OFF) + 3072);                                                              \
        P3_LGKM1(11, X##a4);                                                                          \
        P3_MF(c40, X##a4, X##b0);                                                                     \
        P3_MF(c41, X##a4, X##b1);                                                                     \
        P3_DSR(Y##a4, AY, (OFF) + 4096);                                                              \
        L1;                                                                                           \
        P3_MF(c42, X##a4, X##b2);                                                                     \
        P3_MF(c43, X##a4, X##b3);                                                                     \
        P3_DSR(Y##a5, AY, (OFF) + 5120);                                                              \
        P3_LGKM1(12, X##a5);                                                                          \
        P3_MF(c50, X##a5, X##b0);                                                                     \
        P3_MF(c51, X##a5, X##b1);                                                                     \
        P3_DSR(Y##a6, AY, (OFF) + 6144);                                                              \
        L2;                                                                                           \
        P3_MF(c52, X##a5, X##b2);                                                                     \
        P3_MF(c53, X##a5, X##b3);                                                                     \
        P3_DSR(Y##a7, AY, (OFF) + 7168);                                                              \
        P3_LGKM1(13, X##a6);                                                                          \
        P3_MF(c60, X##a6, X##b0);                                                                     \
        P3_MF(c61, X##a6, X##b1);                                                                     \
        L3;                                                                                           \
        P3_MF(c62, X##a6, X##b2);                                                                     \
        P3_MF(c63, X##a6, X##b3);                                                                     \
        P3_LGKM1(12, X##a7);                                                                          \
        P3_MF(c70, X##a7, X##b0);                                                                     \
        P3_MF(c71, X##a7, X##b1);                                                                     \
        P3_MF(c72, X##a7, X##b2);                                                                     \
        P3_MF(c73, X##a7, X##b3);                                                                     \
    } while (0)
        // interval n: multiply ring slot n & 3, read slot (n + 1) & 3, DMA half-stage n + 3 (planes PA x PB of the
        // chunk at byte offset VO) into slot (n + 3) & 3
#define P3_IV0(VO, PA, PB)                                                                                        \
    P3_STEP(p, q, bA0, bB0, 32768, asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory"), P3_DL(VO, a0p##PA, 3, 0), \
            P3_DL(VO, a1p##PA, 3, 1), P3_DL(VO, b0p##PB, 3, 2), P3_DL(VO, b1p##PB, 3, 3))
#define P3_IV1(VO, PA, PB)                                                                                        \
    P3_STEP(q, p, bA1, bB1, 0, asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory"), P3_DL(VO, a0p##PA, 0, 0),    \
            P3_DL(VO, a1p##PA, 0, 1), P3_DL(VO, b0p##PB, 0, 2), P3_DL(VO, b1p##PB, 0, 3))
#define P3_IV2(VO, PA, PB)                                                                                        \
    P3_STEP(p, q, bA1, bB1, 32768, asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory"), P3_DL(VO, a0p##PA, 1, 0), \
            P3_DL(VO, a1p##PA, 1, 1), P3_DL(VO, b0p##PB, 1, 2), P3_DL(VO, b1p##PB, 1, 3))
#define P3_IV3(VO, PA, PB)                                                                                        \
    P3_STEP(q, p, bA0, bB0, 0, asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory"), P3_DL(VO, a0p##PA, 2, 0),    \
            P3_DL(VO, a1p##PA, 2, 1), P3_DL(VO, b0p##PB, 2, 2), P3_DL(VO, b1p##PB, 2, 3))

        // half-stage order inside a chunk (smallest products first):
        //   NP = 3: (a1,b3) (a3,b1) (a2,b2) (a1,b2) (a2,b1) (a1,b1)      NP = 2: (a1,b2) (a2,b1) (a1,b1)
        int nbody;
        if constexpr (NP == 3) {
            P3_DL4(V0, 0, 2, 0); P3_DL4(V0, 2, 0, 1); P3_DL4(V0, 1, 1, 2);
            nbody = (ce - cb) >> 1;  // 12 half-stages = 2 chunks per loop body
        } else {
            P3_DL4(V0, 0, 1, 0); P3_DL4(V0, 1, 0, 1); P3_DL4(V0, 0, 0, 2);
            nbody = (ce - cb) >> 2;  // 12 half-stages = 4 chunks per loop body
        }
        asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        P3_DSR(pa0, bA0, 0); P3_DSR(pb0, bB0, 0); P3_DSR(pb1, bB0, 1024); P3_DSR(pb2, bB0, 2048); P3_DSR(pb3, bB0, 3072);
        P3_DSR(pa1, bA0, 1024); P3_DSR(pa2, bA0, 2048); P3_DSR(pa3, bA0, 3072); P3_DSR(pa4, bA0, 4096);
        P3_DSR(pa5, bA0, 5120); P3_DSR(pa6, bA0, 6144); P3_DSR(pa7, bA0, 7168);
        for (int it = 0; it < nbody; ++it) {
            if constexpr (NP == 3) {
                P3_IV0(V0, 0, 1); P3_IV1(V0, 1, 0); P3_IV2(V0, 0, 0); P3_IV3(V1, 0, 2);
                P3_IV0(V1, 2, 0); P3_IV1(V1, 1, 1); P3_IV2(V1, 0, 1); P3_IV3(V1, 1, 0);
                P3_IV0(V1, 0, 0); P3_IV1(V2, 0, 2); P3_IV2(V2, 2, 0); P3_IV3(V2, 1, 1);
                V0 = V2;
            } else {
                P3_IV0(V1, 0, 1); P3_IV1(V1, 1, 0); P3_IV2(V1, 0, 0); P3_IV3(V2, 0, 1);
                P3_IV0(V2, 1, 0); P3_IV1(V2, 0, 0); P3_IV2(V3, 0, 1); P3_IV3(V3, 1, 0);
                P3_IV0(V3, 0, 0); P3_IV1(V4, 0, 1); P3_IV2(V4, 1, 0); P3_IV3(V4, 0, 0);
                V0 = V4;
            }
            P3_VNEXT();
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#undef P3_VNEXT
#undef P3_DL
#undef P3_DL4
#undef P3_DSR
#undef P3_LGKM1
#undef P3_LGKM2
#undef P3_MF
#undef P3_STEP
#undef P3_IV0
#undef P3_IV1
#undef P3_IV2
#undef P3_IV3
        // ---- epilogue.  Accumulator c<i><j>[e]: row wm*128 + i*16 + 4*lk + e, column wn*64 + j*16 + lr of the tile
        const int r0 = wm * 128 + 4 * lk, q0 = wn * 64 + lr;
        if (w3) {
            float* __restrict__ dst = P.partial + (size_t)(w3 - 1) * (TILE * TILE);
#define P3_ST(c, i, j)                                                          \
    do {                                                                        \
        float* d_ = dst + (r0 + (i) * 16) * TILE + q0 + (j) * 16;               \
        d_[0] = c[0]; d_[TILE] = c[1]; d_[2 * TILE] = c[2]; d_[3 * TILE] = c[3]; \
    } while (0)
#define P3_ROW(M_, i) M_(c##i##0, i, 0); M_(c##i##1, i, 1); M_(c##i##2, i, 2); M_(c##i##3, i, 3)
            P3_ROW(P3_ST, 0); P3_ROW(P3_ST, 1); P3_ROW(P3_ST, 2); P3_ROW(P3_ST, 3);
            P3_ROW(P3_ST, 4); P3_ROW(P3_ST, 5); P3_ROW(P3_ST, 6); P3_ROW(P3_ST, 7);
#undef P3_ST
        } else {
            float* __restrict__ Cm = P.C;
            const int64_t ldc = P.ldc;
            const int64_t grow = (int64_t)tm * TILE + r0, gcol = (int64_t)tn * TILE + q0;
            const int mode = P.mode;
            const float* rs = NP == 2 ? P.row_scale : nullptr;
            const float* cs = NP == 2 ? P.col_scale : nullptr;
            float cj[4] = {1.f, 1.f, 1.f, 1.f};
            if (rs) {
#pragma unroll
                for (int j = 0; j < 4; ++j) cj[j] = cs[gcol + j * 16];
            }
            const float sgn = mode == 1 ? 1.0f : -1.0f;
            // one row group (16 rows x 64 columns of the wave tile) at a time: all 16 loads of C in flight before
            // the first store (MODE 0), the products scaled (fp16 images) and signed on the way out
#define P3_SC(c, i, j)                                                                             \
    do {                                                                                           \
        if (rs) {                                                                                  \
            const int64_t row_ = grow + (i) * 16;                                                  \
            c[0] *= rs[row_] * cj[j]; c[1] *= rs[row_ + 1] * cj[j];                                \
            c[2] *= rs[row_ + 2] * cj[j]; c[3] *= rs[row_ + 3] * cj[j];                            \
        }                                                                                          \
    } while (0)
#define P3_LDC(t, i, j)                                                                            \
    do {                                                                                           \
        const float* s_ = Cm + (grow + (i) * 16) * ldc + gcol + (j) * 16;                          \
        t[0] = s_[0]; t[1] = s_[ldc]; t[2] = s_[2 * ldc]; t[3] = s_[3 * ldc];                      \
    } while (0)
#define P3_STC(c, i, j)                                                                            \
    do {                                                                                           \
        float* d_ = Cm + (grow + (i) * 16) * ldc + gcol + (j) * 16;                                \
        d_[0] = c[0]; d_[ldc] = c[1]; d_[2 * ldc] = c[2]; d_[3 * ldc] = c[3];                      \
    } while (0)
#define P3_GROUP(i)                                                                                \
    do {                                                                                           \
        P3_SC(c##i##0, i, 0); P3_SC(c##i##1, i, 1); P3_SC(c##i##2, i, 2); P3_SC(c##i##3, i, 3);    \
        if (mode == 0) {                                                                           \
            f32x4 t0, t1, t2, t3;                                                                  \
            P3_LDC(t0, i, 0); P3_LDC(t1, i, 1); P3_LDC(t2, i, 2); P3_LDC(t3, i, 3);                \
            c##i##0 = t0 - c##i##0; c##i##1 = t1 - c##i##1; c##i##2 = t2 - c##i##2; c##i##3 = t3 - c##i##3; \
        } else {                                                                                   \
            c##i##0 *= sgn; c##i##1 *= sgn; c##i##2 *= sgn; c##i##3 *= sgn;                        \
        }                                                                                          \
        P3_STC(c##i##0, i, 0); P3_STC(c##i##1, i, 1); P3_STC(c##i##2, i, 2); P3_STC(c##i##3, i, 3); \
    } while (0)
            P3_GROUP(0); P3_GROUP(1); P3_GROUP(2); P3_GROUP(3); P3_GROUP(4); P3_GROUP(5); P3_GROUP(6); P3_GROUP(7);
#undef P3_SC
#undef P3_LDC
#undef P3_STC
#undef P3_GROUP
#undef P3_ROW
        }
    }
}

// K-split tiles: P = part_0 + part_1 + ... in index order (deterministic), then the tile's epilogue.
// list[2 i] = prob << 28 | tm << 14 | tn, list[2 i + 1] = first slot << 8 | nparts.  Grid (tiles, 16).
template <int NP>
__global__ __launch_bounds__(256) void reduce_kernel(const Group g, const uint32_t* __restrict__ list) {
    const uint32_t w0 = list[2 * blockIdx.x], w1 = list[2 * blockIdx.x + 1];
    const Problem& P = g.p[w0 >> 28];
    const int64_t tm = (w0 >> 14) & 0x3fff, tn = w0 & 0x3fff;
    const int np = (int)(w1 & 0xff);
    const float* __restrict__ part = P.partial + (size_t)(w1 >> 8) * (TILE * TILE);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = (int)threadIdx.x + q * 256, r = (int)blockIdx.y * 16 + idx / 64, c4 = (idx % 64) * 4;
        float4 a = *reinterpret_cast<const float4*>(part + r * TILE + c4);
        for (int k = 1; k < np; ++k) {
            const float4 b = *reinterpret_cast<const float4*>(part + (size_t)k * (TILE * TILE) + r * TILE + c4);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        const int64_t row = tm * TILE + r, col = tn * TILE + c4;
        if (NP == 2 && P.row_scale) {
            const float s = P.row_scale[row];
            a.x *= s * P.col_scale[col]; a.y *= s * P.col_scale[col + 1];
            a.z *= s * P.col_scale[col + 2]; a.w *= s * P.col_scale[col + 3];
        }
        float4* cp = reinterpret_cast<float4*>(P.C + row * P.ldc + col);
        if (P.mode == 0) {
            float4 c = *cp;
            c.x -= a.x; c.y -= a.y; c.z -= a.z; c.w -= a.w;
            *cp = c;
        } else if (P.mode == 1) {
            *cp = a;
        } else {
            *cp = make_float4(-a.x, -a.y, -a.z, -a.w);
        }
    }
}

// ------------------------------------------------------------------------------------------------ host: schedule
// kr: k-range of a tile in IMAGE chunks, always [0, len):  0: KC;  1: 8 (tn + 1)  (B^T lower-triangular);
// 2: KC - 8 tn (B lower-triangular, imaged in reverse);  3: 8 (tm + 1) (A lower-triangular).  lower: tm >= tn only.
struct GemmShape {
    int MT, NT, KC, kr;
    bool lower;
};
struct Plan {
    std::vector<uint32_t> table;    // what the kernel reads (counters, queue bounds, units)
    std::vector<uint32_t> rlist;    // reduce list (2 words per split tile)
    int nslots = 0;
    double makespan = 0, ideal = 0;  // in chunks per workgroup (diagnostics)
};
inline int tile_len(const GemmShape& s, int tm, int tn) {
    int len = s.KC;
    if (s.kr == 1) len = std::min(s.KC, CPT * (tn + 1));
    else if (s.kr == 2) len = s.KC - CPT * tn;
    else if (s.kr == 3) len = std::min(s.KC, CPT * (tm + 1));
    return len;
}
// Schedule.  Tiles in super-tile order (8 x 4 blocks sharing 12 operand panels, longest blocks first) form ONE sequence;
// it is cut into 8 contiguous segments of equal work, one per XCD (a cut may fall inside a tile: K-split).  Inside an
// XCD the 32 workgroups take the segment's units round-robin -- so the 32 units running at a time are neighbours in
// the sequence -- for as many whole rounds as stay below the XCD's mean load T; what is left is dealt by the
// wrap-around rule (fill workgroup after workgroup up to T, cutting a tile along k where it overflows): every
// workgroup ends within a few chunks of T, with at most ~33 cut tiles per XCD.  Unit lengths are multiples of `gran`
// chunks (NP = 3: 2, NP = 2: 4; every k boundary is a multiple of 128 = 4 chunks); the cost of a unit is its length plus
// OVH chunks (prologue + epilogue).
inline Plan make_plan(const std::vector<GemmShape>& shapes, int gran, int max_slots) {
    constexpr int OVH = 2, MINP = 8;
    const int slot_base = 0;
    struct Tile { int prob, tm, tn, len; };
    std::vector<Tile> tiles;
    struct Blk { int neg_len, prob, bi, bj; };
    std::vector<Blk> blocks;
    for (int prob = 0; prob < (int)shapes.size(); ++prob) {
        const GemmShape& s = shapes[prob];
        const bool by_m = s.kr == 3;
        const int bm = by_m ? 4 : 8, bn = by_m ? 8 : 4;
        const int nbm = (s.MT + bm - 1) / bm, nbn = (s.NT + bn - 1) / bn;
        for (int bi = 0; bi < nbm; ++bi)
            for (int bj = 0; bj < nbn; ++bj) {
                int mx = 0;
                for (int tm = bi * bm; tm < std::min(s.MT, (bi + 1) * bm); ++tm)
                    for (int tn = bj * bn; tn < std::min(s.NT, (bj + 1) * bn); ++tn)
                        if (!s.lower || tm >= tn) mx = std::max(mx, tile_len(s, tm, tn));
                if (mx > 0) blocks.push_back({-mx, prob, bi, bj});
            }
    }
    std::stable_sort(blocks.begin(), blocks.end(), [](const Blk& a, const Blk& b) { return a.neg_len < b.neg_len; });
    for (auto& b : blocks) {
        const GemmShape& s = shapes[b.prob];
        const bool by_m = s.kr == 3;
        const int bm = by_m ? 4 : 8, bn = by_m ? 8 : 4;
        for (int tm = b.bi * bm; tm < std::min(s.MT, (b.bi + 1) * bm); ++tm)
            for (int tn = b.bj * bn; tn < std::min(s.NT, (b.bj + 1) * bn); ++tn)
                if (!s.lower || tm >= tn) {
                    const int len = tile_len(s, tm, tn);
                    if (len > 0) tiles.push_back({b.prob, tm, tn, len});
                }
    }
    struct Piece { int tile, cb, ce; };
    long total = 0;
    for (auto& t : tiles) total += t.len + OVH;
    // 1. XCD segments
    std::vector<Piece> seg[8];
    {
        long acc = 0;
        int x = 0;
        for (int ti = 0; ti < (int)tiles.size(); ++ti) {
            int at = 0;
            const int len = tiles[ti].len;
            while (at < len) {
                const long end_x = (x == 7) ? (long)1 << 60 : (total * (x + 1)) / 8;
                long room = end_x - acc - OVH;  // chunks of this tile the segment still takes
                int take = len - at;
                if (room < take) {
                    int r = (int)std::max<long>(room, 0) / gran * gran;
                    if (r < MINP) r = 0;                      // too short a head: the whole rest goes to the next XCD
                    if (take - r < MINP) r = take;            // too short a tail: keep it here
                    take = r;
                }
                if (take > 0) {
                    seg[x].push_back({ti, at, at + take});
                    acc += take + OVH;
                    at += take;
                }
                if (at < len || acc >= end_x) {
                    if (x < 7) ++x;
                    else if (take == 0) { seg[x].push_back({ti, at, len}); acc += len - at + OVH; at = len; }
                }
            }
        }
    }
    // 2. per XCD: whole rounds, then wrap-around; T grows from the mean load until everything fits
    std::vector<Piece> wg[NWG];
    double mk = 0;
    for (int x = 0; x < 8; ++x) {
        long tot = 0;
        for (auto& p : seg[x]) tot += p.ce - p.cb + OVH;
        std::vector<Piece> mine[32];
        long load[32];
        for (long T = (tot + 31) / 32;; T += gran) {
            for (int w = 0; w < 32; ++w) { load[w] = 0; mine[w].clear(); }
            size_t done = 0;
            while (done + 32 <= seg[x].size()) {
                bool ok = true;
                for (int w = 0; w < 32 && ok; ++w) {
                    const Piece& p = seg[x][done + w];
                    ok = load[w] + (p.ce - p.cb) + OVH <= T;
                }
                if (!ok) break;
                for (int w = 0; w < 32; ++w) {
                    const Piece& p = seg[x][done + w];
                    load[w] += p.ce - p.cb + OVH;
                    mine[w].push_back(p);
                }
                done += 32;
            }
            int w = 0;
            bool fits = true;
            for (size_t i = done; i < seg[x].size() && fits; ++i) {
                Piece p = seg[x][i];
                while (p.cb < p.ce) {
                    if (w > 31) { fits = false; break; }
                    const int len = p.ce - p.cb;
                    const long room = T - load[w] - OVH;
                    int take;
                    if (room >= len) take = len;
                    else {
                        take = (int)(std::max<long>(room, 0) / gran * gran);
                        if (len - take < MINP) take = (len - MINP) / gran * gran;  // never leave a stub
                        if (take < MINP) { ++w; continue; }                        // nor start with one
                    }
                    mine[w].push_back({p.tile, p.cb, p.cb + take});
                    load[w] += take + OVH;
                    p.cb += take;
                }
            }
            if (fits) break;
        }
        for (int w = 0; w < 32; ++w) {
            wg[w * 8 + x] = mine[w];
            mk = std::max(mk, (double)load[w]);
        }
    }
    // 3. slots for the tiles that ended up in more than one piece (pieces of a tile in k order), the table
    std::vector<int> npieces(tiles.size(), 0), first_slot(tiles.size(), -1), seen(tiles.size(), 0);
    for (int b = 0; b < NWG; ++b)
        for (auto& p : wg[b]) ++npieces[p.tile];
    Plan pl;
    int slots = 0;
    for (size_t t = 0; t < tiles.size(); ++t)
        if (npieces[t] > 1) {
            first_slot[t] = slot_base + slots;
            slots += npieces[t];
            pl.rlist.push_back((uint32_t)tiles[t].prob << 28 | (uint32_t)tiles[t].tm << 14 | (uint32_t)tiles[t].tn);
            pl.rlist.push_back((uint32_t)first_slot[t] << 8 | (uint32_t)npieces[t]);
        }
    pl.nslots = slots;
    pl.table.assign(T_UNITS, 0u);
    unsigned at = 0;
    for (int b = 0; b < NWG; ++b) {
        pl.table[b] = at;
        for (auto& p : wg[b]) {
            const Tile& t = tiles[p.tile];
            pl.table.push_back((uint32_t)t.prob << 28 | (uint32_t)t.tm << 14 | (uint32_t)t.tn);
            pl.table.push_back((uint32_t)p.cb);
            pl.table.push_back((uint32_t)p.ce);
            pl.table.push_back(npieces[p.tile] > 1 ? (uint32_t)(first_slot[p.tile] + seen[p.tile]++ + 1) : 0u);
            ++at;
        }
    }
    pl.table[NWG] = at;
    pl.makespan = mk;
    pl.ideal = (double)total / NWG;
    if (slots > max_slots) pl.nslots = -1;  // the caller refuses (never at the sizes of this path: <= ~35 cut tiles per XCD)
    return pl;
}

template <int NP>
inline int launch_split(bool trans, const float* S, int64_t ld, int64_t R, int64_t K, int tri, int rev, unsigned char* img,
                        unsigned* rmax, float* inv_scale, hipStream_t st) {
    const dim3 grid((unsigned)(K / 32), (unsigned)(R / 128)), block(256);
    ProfScope ps(PT_CHOL_SPLIT, st);
    if constexpr (NP == 2) {
        GQ_HIP(hipMemsetAsync(rmax, 0, (size_t)R * 4, st));
        const dim3 g2((unsigned)(K / 128), (unsigned)(R / 128));
        if (trans) hipLaunchKernelGGL(row_absmax_kernel<true>, g2, block, 0, st, S, ld, tri, rmax);
        else hipLaunchKernelGGL(row_absmax_kernel<false>, g2, block, 0, st, S, ld, tri, rmax);
        GQ_LAUNCH_CHECK();
    }
    if (trans) hipLaunchKernelGGL((split_kernel<NP, true>), grid, block, 0, st, S, ld, (int)(K / 32), tri, rev, img, rmax, inv_scale);
    else hipLaunchKernelGGL((split_kernel<NP, false>), grid, block, 0, st, S, ld, (int)(K / 32), tri, rev, img, rmax, inv_scale);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

template <int NP>
inline int launch_gemm(const Group& g, int n_reduce, const uint32_t* rlist_dev, hipStream_t st) {
    static std::atomic<bool> attr_set{false};
    if (!attr_set) {
        GQ_HIP(hipFuncSetAttribute((const void*)gemm_kernel<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set = true;
    }
    ProfScope ps(PT_CHOL_IMG_GEMM, st);
    hipLaunchKernelGGL(gemm_kernel<NP>, dim3(256), dim3(512), LDS_BYTES, st, g);
    GQ_LAUNCH_CHECK();
    if (n_reduce > 0) {
        hipLaunchKernelGGL(reduce_kernel<NP>, dim3((unsigned)n_reduce, 16), dim3(256), 0, st, g, rlist_dev);
        GQ_LAUNCH_CHECK();
    }
    return GQ_OK;
}

}  // namespace p3
}  // namespace gq
