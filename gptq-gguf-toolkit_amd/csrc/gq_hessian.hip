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
#include <atomic>
#include <stdlib.h>

#include "gq_common.hpp"
#include <utility>
#include <vector>

namespace gq {



typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------- transpose
// X[T,C] (16-bit) -> Xb[C/128 panels][Tp/64 stages][128 rows][64 k]: each (panel, stage) block is
// the 16 KiB LDS image of one SYRK operand stage, contiguous in HBM and ALREADY XOR-swizzled
// (16-byte chunk kc of row r is stored at chunk kc ^ ((r >> 1) & 7)), so the SYRK kernel streams it
// with lane-linear global_load_lds: every wave instruction copies 1 KiB of consecutive bytes.
// Tokens t >= T are zero padding.  One workgroup = one block; 16-byte global accesses on both sides.
__global__ __launch_bounds__(256) void transpose16_kernel(const uint16_t* __restrict__ X, int64_t T, int64_t C,
                                                          uint16_t* __restrict__ Xb, int64_t nstage, int half_mode) {
    // half_mode (syrk16_256e_kernel): Xb[panel][Tp/32 half-stages][128 rows][32 k] (8 KiB blocks, 16-byte chunk
    // kc of row r at chunk kc ^ f((r >> 2) & 3), f = 0,2,3,1)
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
        int p = tid + u * 256, r, kc;
        if (half_mode) {  // p = h*512 + r*4 + c': half-stage h, stored chunk c' holds k-chunk 4h + (c' ^ ((r>>2)&3))
            const int h = p >> 9, q = p & 511;
            r = q >> 2;
            kc = 4 * h + ((q & 3) ^ ((0x78 >> (2 * ((r >> 2) & 3))) & 3));
        } else {
            r = p >> 3;
            kc = (p & 7) ^ ((r >> 1) & 7);  // stored chunk p holds k-chunk kc
        }
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
    // syrk16_256n_kernel only: X given as separate blocks of hs_per_seg * 32 tokens each (the per-sample activation
    // tensors of the forward hooks, read where they lie); segs[k] = device address of block k.  0: Xt is contiguous.
    const uint64_t* segs;
    int hs_per_seg;
};

// scalar load that never overlaps the hand-counted lgkmcnt waits of the SYRK pipeline: issued and waited in one go
__device__ __forceinline__ uint64_t sload64_now(const uint64_t* p) {
    uint64_t v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v) : "s"(p) : "memory");
    return v;
}
struct SyrkGroup {
    int n, total_tiles;
    SyrkProblem p[H_MAX_GROUP];
    // syrk16_256e_kernel only: balanced tile table, entry = problem << 24 | ti << 12 | tj (0xffffffff: none),
    // row x holds the tiles workgroups with blockIdx % 8 == x (= XCD x) process, in order
    const uint32_t* table;
    int per_xcd;
    // syrk16_256n_kernel only (nullptr: none): K-split of the tiles of the last, partial round.  aux[i] belongs to
    // table[i]: 0xffffffff = the whole token range, H updated in the epilogue; else slot << 12 | part << 6 | nparts
    // = tokens [part, part+1) * T / nparts (in units of 128), raw fp32 sums stored to partial[slot] (256 x 256)
    const uint32_t* aux;
    float* partial;
    // syrk16_256n_kernel, persistent launch (one workgroup per CU walks its XCD's list): bar[x] counts the workgroups
    // of XCD x that finished a round; a round starts when all of them have, so the 32 tiles an XCD works on at a
    // time stay in step and share their 12 operand panels through the XCD's L2.  nullptr: one tile per workgroup.
    unsigned* bar;
    // syrk16_256w_kernel (r05): soft XCD rendezvous INSIDE a tile, every `ck` half-stages (a power of two; 0: none).  Between
    // the round-start rendezvous the 32 workgroups of an XCD drift apart by more than the ~20 half-stages of panels their 4 MB
    // L2 holds, and a panel byte that should be fetched once per XCD is fetched again from the fabric: the kernel is bound by
    // exactly that delivery (profiles/r05_syrk_decompose.txt; zeros, C = 14336: 1.34 -> 1.76 PFLOP/s with ck = 256).  bar[8 + x]
    // counts checkpoint arrivals of XCD x; every workgroup contributes exactly `nck` per round (the checkpoints it passes, the
    // rest when its unit ends, all of them when it has no unit), so checkpoint k of round r waits for (r nck + k) x workgroups
    // -- at most ~4 us: peers that are not resident can never hang it, and results do not depend on it.
    int ck, nck;
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

    // operand stage (panel, s) is the contiguous, pre-swizzled 16 KiB block Xb[(panel*nstage + s)*8192 ...];
    // thread tid owns chunks p = u*256 + tid (u = 0..3) of both operands: global -> VGPR -> LDS, linear.
    // Register prefetch runs TWO stages ahead (hipcc counts vmcnt per register, so only the stage being
    // written to LDS is waited for; with global_load_lds it drains everything before any ds_read).
    const int64_t nk = Tp / HK;
    const uint4* srcA = reinterpret_cast<const uint4*>(Xt + (ti * nk) * (HT * HK)) + tid;
    const uint4* srcB = reinterpret_cast<const uint4*>(Xt + (tj * nk) * (HT * HK)) + tid;
    // two register sets with static names (arrays indexed through references end up in scratch)
    uint4 ea0, ea1, ea2, ea3, eb0, eb1, eb2, eb3;  // "even" set
    uint4 oa0, oa1, oa2, oa3, ob0, ob1, ob2, ob3;  // "odd" set
#define GQ_FETCH(P, s)                                                                             \
    do {                                                                                           \
        const uint4* pa_ = srcA + (s) * 1024;                                                      \
        const uint4* pb_ = srcB + (s) * 1024;                                                      \
        P##a0 = pa_[0]; P##a1 = pa_[256]; P##a2 = pa_[512]; P##a3 = pa_[768];                      \
        P##b0 = pb_[0]; P##b1 = pb_[256]; P##b2 = pb_[512]; P##b3 = pb_[768];                      \
    } while (0)
#define GQ_COMMIT(P, buf)                                                                          \
    do {                                                                                           \
        uint4* la_ = reinterpret_cast<uint4*>(smem + (buf) * H_STAGE_BYTES) + tid;                 \
        la_[0] = P##a0; la_[256] = P##a1; la_[512] = P##a2; la_[768] = P##a3;                      \
        la_[1024] = P##b0; la_[1280] = P##b1; la_[1536] = P##b2; la_[1792] = P##b3;                \
    } while (0)
    const int li = lane & 31, lk = lane >> 5;
    int offA[2], offB[2], swz[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra_ = wm * 64 + i * 32 + li, rb_ = wn * 64 + i * 32 + li;
        offA[i] = ra_ * 128;
        offB[i] = HT * HK * 2 + rb_ * 128;
        swz[0][i] = (ra_ >> 1) & 7;
        swz[1][i] = (rb_ >> 1) & 7;
    }
    auto compute = [&](int buf) {
        const unsigned char* base = smem + buf * H_STAGE_BYTES;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int kc = s4 * 2 + lk;
            uint4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const uint4*>(base + offA[i] + ((kc ^ swz[0][i]) << 4));
                b[i] = *reinterpret_cast<const uint4*>(base + offB[i] + ((kc ^ swz[1][i]) << 4));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma16<BF16>(a[i], b[j], acc[i][j]);
        }
    };
    // prologue: stage 0 -> LDS buf 0, stage 1 -> odd register set
    GQ_FETCH(e, 0);
    GQ_COMMIT(e, 0);
    if (nk > 1) GQ_FETCH(o, 1);
    __syncthreads();
    for (int64_t t = 0; t < nk; t += 2) {
        // even step: compute stage t (buf 0); stage t+1 sits in the odd set, stage t+2 goes to the even set
        if (t + 2 < nk) GQ_FETCH(e, t + 2);
        compute(0);
        if (t + 1 < nk) GQ_COMMIT(o, 1);
        __syncthreads();
        if (t + 1 < nk) {
            // odd step: compute stage t+1 (buf 1); stage t+2 sits in the even set, stage t+3 goes to the odd set
            if (t + 3 < nk) GQ_FETCH(o, t + 3);
            compute(1);
            if (t + 2 < nk) GQ_COMMIT(e, 0);
            __syncthreads();
        }
    }
#undef GQ_FETCH
#undef GQ_COMMIT
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

// ------------------------------------------ 16-bit SYRK, 256x256 tiles, direct-to-LDS operand ring
constexpr int BT = 256;                         // tile
constexpr int SK = 32;                          // k per half-stage
constexpr int S_BUF_BYTES = 2 * BT * SK * 2;    // one ring slot: A 16 KiB | B 16 KiB
constexpr int S_NBUF = 4;
constexpr int S_LDS_BYTES = S_NBUF * S_BUF_BYTES;  // 128 KiB
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// profiles/micro/mfma_power.hip: with random fp16 operands the chip is POWER-limited on MFMA work -- a
// register-resident loop sustains 1.70 PFLOP/s with v_mfma_f32_32x32x16_f16 and 1.95 PFLOP/s with
// v_mfma_f32_16x16x32_f16 at two waves per SIMD (2.48 PFLOP/s on all-zero data), and the SYRK kernels run
// at 1.3-1.45 GHz -- hence the cheaper instruction.  8 waves (2 x 4), wave tile 128 x 64 = 8 x 4
// accumulators of 16x16 (128 VGPRs), one workgroup per CU.  K advances in half-stages of 32 (one 32 KiB LDS
// image: A 256 rows x 64 B | B 256 rows x 64 B) through a ring of FOUR slots filled by
// global_load_lds_dwordx4 (no VGPR round trip, no ds_write).  The compiler serialises every ds_read behind
// vmcnt(0) once an LDS-DMA load is in flight, so the main loop is inline asm in exactly the written order
// and the counters are managed by hand.  One ds_read_b128 returns a whole 16-row x 32-k
// fragment (lane = row & 15, k-chunk = lane >> 4), so a k32 half-stage is ONE step of 32 MFMAs with the
// 12 fragment reads of the next half-stage behind MFMAs 1,3,...,23, the barrier behind MFMA 0 and the 4 DMA
// pieces of half-stage n+3 behind MFMAs 13/17/21/25.  At barrier n the loads younger than half-stage
// n+1's are the 4 of half-stage n+2 -> vmcnt(4).  Operand image: transpose16_kernel half_mode (chunk kc of
// row r at kc ^ f((r >> 2) & 3), f = 0,2,3,1: conflict-free for this fragment shape).
template <bool BF16>
__global__ __launch_bounds__(512, 2) void syrk16_256e_kernel(const SyrkGroup grp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    // Workgroup b serves XCD b % 8 and walks that XCD's tile list with stride gridDim.x / 8 (one tile each with the
    // default grid).  Measured with a capped grid (64-192 workgroups) for the gap-filling second launch of a
    // block: no gain over the uncapped launch -- the step is bound by total work, not by the launch order.
    for (int slot = (int)(blockIdx.x >> 3); slot < grp.per_xcd; slot += (int)(gridDim.x >> 3)) {
    // readfirstlane: everything derived from the entry (panel base pointers used as SGPR asm operands) must be
    // provably wave-uniform for the compiler
    const uint32_t ent = (uint32_t)__builtin_amdgcn_readfirstlane((int)grp.table[(blockIdx.x & 7) * grp.per_xcd + slot]);
    if (ent == 0xffffffffu) break;  // lists are dense; padding only at the end
    __syncthreads();                // every wave is done with the ring of the previous tile
    const SyrkProblem& P = grp.p[ent >> 24];
    const int64_t ti = (ent >> 12) & 0xfff, tj = ent & 0xfff;
    const int64_t C = P.C;
    const int nhs = (int)(P.Tp / SK);
    const uint16_t* __restrict__ Xt = P.Xt;
    // wave-uniform panel streams (SGPR pairs) + one 32-bit per-lane byte offset that advances 8 KiB per
    // half-stage and is clamped to the last one (keeps the vmcnt arithmetic valid in the tail)
    const char* sp0 = reinterpret_cast<const char*>(Xt + ((2 * ti) * (int64_t)nhs) * (HT * SK));
    const char* sp1 = reinterpret_cast<const char*>(Xt + ((2 * ti + 1) * (int64_t)nhs) * (HT * SK));
    const char* sp2 = reinterpret_cast<const char*>(Xt + ((2 * tj) * (int64_t)nhs) * (HT * SK));
    const char* sp3 = reinterpret_cast<const char*>(Xt + ((2 * tj + 1) * (int64_t)nhs) * (HT * SK));
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const unsigned vlast = (unsigned)(nhs - 1) * 8192u + (unsigned)tid * 16u;
    unsigned voff = (unsigned)tid * 16u;
    const unsigned ldsw = lds0 + (unsigned)wid * 1024u;  // + slot * 32 KiB + piece * 8 KiB; hardware adds lane * 16
#define GQ_EDL(sp, slot, part) GQ_EDL_(sp, slot, part)
#define GQ_EDL_(sp, slot, part)                                                                       \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"                      \
                 :: "v"(voff), "s"(sp), "s"(ldsw + (unsigned)((slot) * S_BUF_BYTES + (part) * 8192)) : "memory")
#define GQ_EADV() voff = voff + 8192u < vlast ? voff + 8192u : vlast
    const int lr = lane & 15, lk = lane >> 4;
    unsigned bA0, bA1, bB0, bB1;  // b<op><+64 KiB>
    {
        const unsigned sw = (0x78u >> (2 * ((lr >> 2) & 3))) & 3u;
        const unsigned ch = ((unsigned)lk ^ sw) << 4;
        bA0 = lds0 + (unsigned)(wm * 128 + lr) * 64u + ch;
        bB0 = lds0 + 16384u + (unsigned)(wn * 64 + lr) * 64u + ch;
        bA1 = bA0 + 65536u;
        bB1 = bB0 + 65536u;
    }
    f32x4 c00, c01, c02, c03, c10, c11, c12, c13, c20, c21, c22, c23, c30, c31, c32, c33, c40, c41, c42, c43, c50, c51, c52, c53, c60, c61, c62, c63, c70, c71, c72, c73;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        c00[e] = 0.f;
        c01[e] = 0.f;
        c02[e] = 0.f;
        c03[e] = 0.f;
        c10[e] = 0.f;
        c11[e] = 0.f;
        c12[e] = 0.f;
        c13[e] = 0.f;
        c20[e] = 0.f;
        c21[e] = 0.f;
        c22[e] = 0.f;
        c23[e] = 0.f;
        c30[e] = 0.f;
        c31[e] = 0.f;
        c32[e] = 0.f;
        c33[e] = 0.f;
        c40[e] = 0.f;
        c41[e] = 0.f;
        c42[e] = 0.f;
        c43[e] = 0.f;
        c50[e] = 0.f;
        c51[e] = 0.f;
        c52[e] = 0.f;
        c53[e] = 0.f;
        c60[e] = 0.f;
        c61[e] = 0.f;
        c62[e] = 0.f;
        c63[e] = 0.f;
        c70[e] = 0.f;
        c71[e] = 0.f;
        c72[e] = 0.f;
        c73[e] = 0.f;
    }
    u32x4 pa0, pa1, pa2, pa3, pa4, pa5, pa6, pa7, pb0, pb1, pb2, pb3;
    u32x4 qa0, qa1, qa2, qa3, qa4, qa5, qa6, qa7, qb0, qb1, qb2, qb3;
#define GQ_EDSR(dst, base, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(off) : "memory")
#define GQ_ELGKM1(N, x) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(x)::"memory")
#define GQ_ELGKM2(N, x, y) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(x), "+v"(y)::"memory")
#define GQ_EMF(c, a, b)                                                                               \
    do {                                                                                              \
        if constexpr (BF16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); \
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));       \
    } while (0)
    // reads of a set are issued a0,b0,b1,b2,b3,a1..a7; lgkmcnt(k) = all but the youngest k LDS reads are back
#define GQ_ESTEP(X, Y, AY, BY, OFF, MID, L0, L1, L2, L3)                                              \
    do {                                                                                              \
        GQ_ELGKM2(10, X##a0, X##b0);                                                                  \
        GQ_EMF(c00, X##a0, X##b0);                                                                    \
        MID;                                                                                          \
        GQ_ELGKM1(9, X##b1);                                                                          \
        GQ_EMF(c01, X##a0, X##b1);                                                                    \
        GQ_EDSR(Y##a0, AY, (OFF) + 0);                                                                \
        GQ_ELGKM1(9, X##b2);                                                                          \
        GQ_EMF(c02, X##a0, X##b2);                                                                    \
        GQ_ELGKM1(8, X##b3);                                                                          \
        GQ_EMF(c03, X##a0, X##b3);                                                                    \
        GQ_EDSR(Y##b0, BY, (OFF) + 0);                                                                \
        GQ_ELGKM1(8, X##a1);                                                                          \
        GQ_EMF(c10, X##a1, X##b0);                                                                    \
        GQ_EMF(c11, X##a1, X##b1);                                                                    \
        GQ_EDSR(Y##b1, BY, (OFF) + 1024);                                                             \
        GQ_EMF(c12, X##a1, X##b2);                                                                    \
        GQ_EMF(c13, X##a1, X##b3);                                                                    \
        GQ_EDSR(Y##b2, BY, (OFF) + 2048);                                                             \
        GQ_ELGKM1(9, X##a2);                                                                          \
        GQ_EMF(c20, X##a2, X##b0);                                                                    \
        GQ_EMF(c21, X##a2, X##b1);                                                                    \
        GQ_EDSR(Y##b3, BY, (OFF) + 3072);                                                             \
        GQ_EMF(c22, X##a2, X##b2);                                                                    \
        GQ_EMF(c23, X##a2, X##b3);                                                                    \
        GQ_EDSR(Y##a1, AY, (OFF) + 1024);                                                             \
        GQ_ELGKM1(10, X##a3);                                                                         \
        GQ_EMF(c30, X##a3, X##b0);                                                                    \
        GQ_EMF(c31, X##a3, X##b1);                                                                    \
        GQ_EDSR(Y##a2, AY, (OFF) + 2048);                                                             \
        L0;                                                                                           \
        GQ_EMF(c32, X##a3, X##b2);                                                                    \
        GQ_EMF(c33, X##a3, X##b3);                                                                    \
        GQ_EDSR(Y##a3, AY, (OFF) + 3072);                                                             \
        GQ_ELGKM1(11, X##a4);                                                                         \
        GQ_EMF(c40, X##a4, X##b0);                                                                    \
        GQ_EMF(c41, X##a4, X##b1);                                                                    \
        GQ_EDSR(Y##a4, AY, (OFF) + 4096);                                                             \
        L1;                                                                                           \
        GQ_EMF(c42, X##a4, X##b2);                                                                    \
        GQ_EMF(c43, X##a4, X##b3);                                                                    \
        GQ_EDSR(Y##a5, AY, (OFF) + 5120);                                                             \
        GQ_ELGKM1(12, X##a5);                                                                         \
        GQ_EMF(c50, X##a5, X##b0);                                                                    \
        GQ_EMF(c51, X##a5, X##b1);                                                                    \
        GQ_EDSR(Y##a6, AY, (OFF) + 6144);                                                             \
        L2;                                                                                           \
        GQ_EMF(c52, X##a5, X##b2);                                                                    \
        GQ_EMF(c53, X##a5, X##b3);                                                                    \
        GQ_EDSR(Y##a7, AY, (OFF) + 7168);                                                             \
        GQ_ELGKM1(13, X##a6);                                                                         \
        GQ_EMF(c60, X##a6, X##b0);                                                                    \
        GQ_EMF(c61, X##a6, X##b1);                                                                    \
        L3;                                                                                           \
        GQ_EMF(c62, X##a6, X##b2);                                                                    \
        GQ_EMF(c63, X##a6, X##b3);                                                                    \
        GQ_ELGKM1(12, X##a7);                                                                         \
        GQ_EMF(c70, X##a7, X##b0);                                                                    \
        GQ_EMF(c71, X##a7, X##b1);                                                                    \
        GQ_EMF(c72, X##a7, X##b2);                                                                    \
        GQ_EMF(c73, X##a7, X##b3);                                                                    \
    } while (0)
#define GQ_EINTERVAL(X, Y, AY, BY, OFF, SLOT3)                                                        \
    GQ_ESTEP(X, Y, AY, BY, OFF, asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory"),          \
             GQ_EDL(sp0, SLOT3, 0), GQ_EDL(sp1, SLOT3, 1), GQ_EDL(sp2, SLOT3, 2), GQ_EDL(sp3, SLOT3, 3)); \
    GQ_EADV();

    for (int h = 0; h < 3; ++h) {
        GQ_EDL_(sp0, h, 0); GQ_EDL_(sp1, h, 1); GQ_EDL_(sp2, h, 2); GQ_EDL_(sp3, h, 3);
        GQ_EADV();
    }
    asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    GQ_EDSR(pa0, bA0, 0); GQ_EDSR(pb0, bB0, 0); GQ_EDSR(pb1, bB0, 1024); GQ_EDSR(pb2, bB0, 2048); GQ_EDSR(pb3, bB0, 3072);
    GQ_EDSR(pa1, bA0, 1024); GQ_EDSR(pa2, bA0, 2048); GQ_EDSR(pa3, bA0, 3072); GQ_EDSR(pa4, bA0, 4096);
    GQ_EDSR(pa5, bA0, 5120); GQ_EDSR(pa6, bA0, 6144); GQ_EDSR(pa7, bA0, 7168);
    for (int n = 0; n < nhs; n += 4) {  // nhs % 4 == 0; interval n: multiply slot n & 3, read slot (n+1) & 3
        GQ_EINTERVAL(p, q, bA0, bB0, 32768, 3)
        GQ_EINTERVAL(q, p, bA1, bB1, 0, 0)
        GQ_EINTERVAL(p, q, bA1, bB1, 32768, 1)
        GQ_EINTERVAL(q, p, bA0, bB0, 0, 2)
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#undef GQ_EDL
#undef GQ_EDL_
#undef GQ_EADV
#undef GQ_EDSR
#undef GQ_ELGKM1
#undef GQ_ELGKM2
#undef GQ_EMF
#undef GQ_ESTEP
#undef GQ_EINTERVAL
    float* __restrict__ H = P.H;
    const float beta = P.beta, alpha = P.alpha;
    const int64_t i0 = ti * BT + wm * 128 + 4 * lk, j0 = tj * BT + wn * 64 + lr;
    // the four accumulator elements of a lane are four consecutive ROWS of one column: the mirrored copy is one
    // 16-byte store per lane (four lanes = one 64-byte segment of the mirrored row)
#define GQ_ESTORE(c, i, j)                                                                            \
    do {                                                                                              \
        const int64_t col = j0 + (j) * 16, row = i0 + (i) * 16;                                       \
        float4 h;                                                                                     \
        h.x = beta * H[(row + 0) * C + col] + alpha * c[0];                                           \
        h.y = beta * H[(row + 1) * C + col] + alpha * c[1];                                           \
        h.z = beta * H[(row + 2) * C + col] + alpha * c[2];                                           \
        h.w = beta * H[(row + 3) * C + col] + alpha * c[3];                                           \
        H[(row + 0) * C + col] = h.x; H[(row + 1) * C + col] = h.y;                                   \
        H[(row + 2) * C + col] = h.z; H[(row + 3) * C + col] = h.w;                                   \
        if (ti != tj) *reinterpret_cast<float4*>(H + col * C + row) = h;                              \
    } while (0)
    GQ_ESTORE(c00, 0, 0);
    GQ_ESTORE(c01, 0, 1);
    GQ_ESTORE(c02, 0, 2);
    GQ_ESTORE(c03, 0, 3);
    GQ_ESTORE(c10, 1, 0);
    GQ_ESTORE(c11, 1, 1);
    GQ_ESTORE(c12, 1, 2);
    GQ_ESTORE(c13, 1, 3);
    GQ_ESTORE(c20, 2, 0);
    GQ_ESTORE(c21, 2, 1);
    GQ_ESTORE(c22, 2, 2);
    GQ_ESTORE(c23, 2, 3);
    GQ_ESTORE(c30, 3, 0);
    GQ_ESTORE(c31, 3, 1);
    GQ_ESTORE(c32, 3, 2);
    GQ_ESTORE(c33, 3, 3);
    GQ_ESTORE(c40, 4, 0);
    GQ_ESTORE(c41, 4, 1);
    GQ_ESTORE(c42, 4, 2);
    GQ_ESTORE(c43, 4, 3);
    GQ_ESTORE(c50, 5, 0);
    GQ_ESTORE(c51, 5, 1);
    GQ_ESTORE(c52, 5, 2);
    GQ_ESTORE(c53, 5, 3);
    GQ_ESTORE(c60, 6, 0);
    GQ_ESTORE(c61, 6, 1);
    GQ_ESTORE(c62, 6, 2);
    GQ_ESTORE(c63, 6, 3);
    GQ_ESTORE(c70, 7, 0);
    GQ_ESTORE(c71, 7, 1);
    GQ_ESTORE(c72, 7, 2);
    GQ_ESTORE(c73, 7, 3);
#undef GQ_ESTORE
    }  // tile loop
}

// -------------- 16-bit SYRK, 256x256 tiles, NATURAL activation layout (no re-layout pass), gfx950 transpose reads
// Same tile / wave shape, ring, hand-ordered stream and tile table as syrk16_256e_kernel, but the ring slots hold
// X as it lies in memory -- [32 tokens][256 channels] per operand, 512-byte rows -- and the MFMA fragments are
// assembled by ds_read_b64_tr_b16: per 16-lane group, lane j passes the address of 4 consecutive channels of token
// row j>>2 and receives channel j of that 4x16 block, i.e. 4 consecutive tokens of one channel; two reads (token
// rows +0..3 and +4..7) are the 8-deep k-slice of one lane of a 16x16x32 operand (profiles/micro/tr_read_probe.hip).
// Bank conflicts: the 8 rows one instruction touches are 512 bytes apart; the 32-byte fragment slot F of row r is
// therefore stored at slot F ^ g(r), g(r) = (r & 3) | ((r >> 3) & 1) << 2 (conflict-free, tr_read_banks.hip) -- the
// permutation is applied by the DMA source addresses, the fragment address is base ^ (i << 5).
// Per k32 step: 32 MFMAs; fragments a0,b0..b3,a1,a2 of the NEXT half-stage are read behind MFMAs 12..25 into the
// other register set, a3..a7 of the CURRENT one behind MFMAs 0..9 (single set) -- never more than 15 LDS reads in
// flight (lgkmcnt is 4 bits).  Requires T % 128 == 0 (whole ring turns); other T take the re-layout path.
// (Measured and removed, r04, profiles/r04_syrk_stagger_ab.txt: waves 4-7 issuing their DMA pieces behind the step's last
// MFMAs instead of its middle -- 1.21 vs 1.24 PFLOP/s, matrix pipe 76.5 vs 80.2 % busy; s_setprio 1 for waves 4-7: no change.)
template <bool BF16>
__global__ __launch_bounds__(512, 2) void syrk16_256n_kernel(const SyrkGroup grp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    int round = 0;
    for (int slot = (int)(blockIdx.x >> 3); slot < grp.per_xcd; slot += (int)(gridDim.x >> 3), ++round) {
    if (grp.bar && round > 0) {  // XCD-wide rendezvous between rounds (persistent launch)
        if (tid == 0) {
            unsigned* b = grp.bar + (blockIdx.x & 7);
            __hip_atomic_fetch_add(b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)round * (gridDim.x >> 3);
            // a SOFT rendezvous (performance only, results do not depend on it): give up after ~30 us, so that a
            // workgroup whose peers are not resident (another process on the same GPU) can never wait for ever
            for (int spin = 0; spin < 64 && __hip_atomic_load(b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want; ++spin)
                __builtin_amdgcn_s_sleep(16);
        }
    }
    const uint32_t ent = (uint32_t)__builtin_amdgcn_readfirstlane((int)grp.table[(blockIdx.x & 7) * grp.per_xcd + slot]);
    if (ent == 0xffffffffu) {
        if (grp.bar) continue;  // keeps taking part in the rendezvous
        break;
    }
    const uint32_t aux = grp.aux ? (uint32_t)__builtin_amdgcn_readfirstlane((int)grp.aux[(blockIdx.x & 7) * grp.per_xcd + slot])
                                 : 0xffffffffu;
    __syncthreads();
    const SyrkProblem& P = grp.p[ent >> 24];
    const int64_t ti = (ent >> 12) & 0xfff, tj = ent & 0xfff;
    const int64_t C = P.C;
    // token range of this unit, in units of 128 tokens (one turn of the ring): all of them, or part k of n
    int64_t u0 = 0, u1 = P.Tp / (4 * SK);      // Tp = T here
    if (aux != 0xffffffffu) {
        const int64_t np = aux & 63, kp = (aux >> 6) & 63;
        u0 = kp * u1 / np;
        u1 = (kp + 1) * (P.Tp / (4 * SK)) / np;
    }
    const int nhs = (int)((u1 - u0) * 4);
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    // ---- DMA: waves 0-3 bring the A image (channels 256 ti ..), waves 4-7 the B image; wave w & 3 owns token rows
    // 8 (w & 3) .. +7 of every half-stage as four 1 KiB pieces of two rows each
    const int op = wid >> 2, rb = 8 * (wid & 3);
    const int64_t hstride = 32 * C * 2;        // bytes per half-stage
    const int64_t colofs = (op ? tj : ti) * 512;
    const char* gsrc = reinterpret_cast<const char*>(P.Xt) + colofs + u0 * 4 * hstride;
    // segmented X: half-stage (u0 * 4 + h) lies in block (..) / hs_per_seg at row 32 * ((..) % hs_per_seg)
    int hsps = P.hs_per_seg, seg_within = 0;
    const uint64_t* segp = P.segs;
    const char* segbase = nullptr;
    if (hsps) {
        const int h0 = (int)(u0 * 4);
        segp += h0 / hsps;
        seg_within = h0 % hsps;
        segbase = reinterpret_cast<const char*>(sload64_now(segp)) + colofs;
        gsrc = segbase + (int64_t)seg_within * hstride;
    }
    asm volatile("" : "+s"(hsps), "+s"(segp));  // opaque SGPR values: never re-loaded from the kernel arguments
    unsigned voff0, voff1, voff2, voff3;
    {
        const int hrow = lane >> 5, s16 = lane & 31;
#define GQ_NVOFF(u)                                                                                   \
        ([&] {                                                                                        \
            const int r_ = rb + 2 * (u) + hrow;                                                       \
            const int g_ = (r_ & 3) | (((r_ >> 3) & 1) << 2);                                         \
            return (unsigned)(r_ * C * 2) + (unsigned)((((s16 >> 1) ^ g_) << 5) + ((s16 & 1) << 4));  \
        }())
        voff0 = GQ_NVOFF(0); voff1 = GQ_NVOFF(1); voff2 = GQ_NVOFF(2); voff3 = GQ_NVOFF(3);
#undef GQ_NVOFF
    }
    const unsigned ldsw = lds0 + (unsigned)(op * 16384 + rb * 512);
    int hnext = 0;
    const char* gbase = gsrc;
#define GQ_NDL(vo, slot_, u) GQ_NDL_(vo, slot_, u)
#define GQ_NDL_(vo, slot_, u)                                                                         \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"                      \
                 :: "v"(vo), "s"(gbase), "s"(ldsw + (unsigned)((slot_) * S_BUF_BYTES + (u) * 1024)) : "memory")
#define GQ_NADV()                                                                                     \
    do {                                                                                              \
        if (hnext + 1 < nhs) {                                                                        \
            ++hnext;                                                                                  \
            if (hsps) {                                                                               \
                if (++seg_within == hsps) {                                                           \
                    seg_within = 0;                                                                   \
                    ++segp;                                                                           \
                    segbase = reinterpret_cast<const char*>(sload64_now(segp)) + colofs;              \
                }                                                                                     \
                gbase = segbase + (int64_t)seg_within * hstride;                                      \
            } else {                                                                                  \
                gbase = gsrc + (int64_t)hnext * hstride;                                              \
            }                                                                                         \
        }                                                                                             \
    } while (0)
    // ---- fragment addresses (see header): lane = (k-group kc, row-in-group q, 8-byte chunk ch)
    unsigned bAlo, bAhi, bBlo, bBhi;
    {
        const int kc = lane >> 4, q = (lane & 15) >> 2, ch = lane & 3;
        const int r = 8 * kc + q, g = q | ((kc & 1) << 2);
        bAlo = lds0 + (unsigned)(r * 512 + (((wm * 8) | g) << 5) + ch * 8);
        bBlo = lds0 + 16384u + (unsigned)(r * 512 + ((((wn << 2)) ^ g) << 5) + ch * 8);
        bAhi = bAlo + 65536u;
        bBhi = bBlo + 65536u;
    }
    f32x4 c00, c01, c02, c03, c10, c11, c12, c13, c20, c21, c22, c23, c30, c31, c32, c33, c40, c41, c42, c43, c50, c51, c52, c53, c60, c61, c62, c63, c70, c71, c72, c73;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        c00[e] = 0.f;
        c01[e] = 0.f;
        c02[e] = 0.f;
        c03[e] = 0.f;
        c10[e] = 0.f;
        c11[e] = 0.f;
        c12[e] = 0.f;
        c13[e] = 0.f;
        c20[e] = 0.f;
        c21[e] = 0.f;
        c22[e] = 0.f;
        c23[e] = 0.f;
        c30[e] = 0.f;
        c31[e] = 0.f;
        c32[e] = 0.f;
        c33[e] = 0.f;
        c40[e] = 0.f;
        c41[e] = 0.f;
        c42[e] = 0.f;
        c43[e] = 0.f;
        c50[e] = 0.f;
        c51[e] = 0.f;
        c52[e] = 0.f;
        c53[e] = 0.f;
        c60[e] = 0.f;
        c61[e] = 0.f;
        c62[e] = 0.f;
        c63[e] = 0.f;
        c70[e] = 0.f;
        c71[e] = 0.f;
        c72[e] = 0.f;
        c73[e] = 0.f;
    }
    u32x2 pa0l, pa0h, pb0l, pb0h, pb1l, pb1h, pb2l, pb2h, pb3l, pb3h, pa1l, pa1h, pa2l, pa2h;
    u32x2 qa0l, qa0h, qb0l, qb0h, qb1l, qb1h, qb2l, qb2h, qb3l, qb3h, qa1l, qa1h, qa2l, qa2h;
    u32x2 la3l, la3h, la4l, la4h, la5l, la5h, la6l, la6h, la7l, la7h;
#define GQ_NRD(dst, base, idx, off)                                                                   \
    do {                                                                                              \
        const unsigned t_ = (base) ^ ((unsigned)(idx) << 5);                                          \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(t_), "n"(off) : "memory"); \
    } while (0)
#define GQ_NWAIT(N, x, y) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(x), "+v"(y)::"memory")
#define GQ_NMF(c, al, ah, bl, bh)                                                                     \
    do {                                                                                              \
        const u32x4 A_ = __builtin_shufflevector(al, ah, 0, 1, 2, 3), B_ = __builtin_shufflevector(bl, bh, 0, 1, 2, 3); \
        if constexpr (BF16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(A_), "v"(B_)); \
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(A_), "v"(B_));     \
    } while (0)
#define GQ_NBAR() asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory")
#define GQ_NSTEP(X, Y, AC, OFFC, AN, BN, OFFN, L0, L1, L2, L3)                                        \
    do {                                                                                              \
        GQ_NWAIT(12, X##a0l, X##a0h);                                                                 \
        GQ_NWAIT(10, X##b0l, X##b0h);                                                                 \
        GQ_NMF(c00, X##a0l, X##a0h, X##b0l, X##b0h);                                                  \
        GQ_NRD(la3l, AC, 3, (OFFC) + 0);                                                              \
        GQ_NWAIT(9, X##b1l, X##b1h);                                                                  \
        GQ_NMF(c01, X##a0l, X##a0h, X##b1l, X##b1h);                                                  \
        GQ_NRD(la3h, AC, 3, (OFFC) + 2048);                                                           \
        GQ_NWAIT(8, X##b2l, X##b2h);                                                                  \
        GQ_NMF(c02, X##a0l, X##a0h, X##b2l, X##b2h);                                                  \
        GQ_NRD(la4l, AC, 4, (OFFC) + 0);                                                              \
        GQ_NWAIT(7, X##b3l, X##b3h);                                                                  \
        GQ_NMF(c03, X##a0l, X##a0h, X##b3l, X##b3h);                                                  \
        GQ_NRD(la4h, AC, 4, (OFFC) + 2048);                                                           \
        GQ_NWAIT(6, X##a1l, X##a1h);                                                                  \
        GQ_NMF(c10, X##a1l, X##a1h, X##b0l, X##b0h);                                                  \
        GQ_NRD(la5l, AC, 5, (OFFC) + 0);                                                              \
        GQ_NMF(c11, X##a1l, X##a1h, X##b1l, X##b1h);                                                  \
        GQ_NRD(la5h, AC, 5, (OFFC) + 2048);                                                           \
        GQ_NMF(c12, X##a1l, X##a1h, X##b2l, X##b2h);                                                  \
        GQ_NRD(la6l, AC, 6, (OFFC) + 0);                                                              \
        GQ_NMF(c13, X##a1l, X##a1h, X##b3l, X##b3h);                                                  \
        GQ_NRD(la6h, AC, 6, (OFFC) + 2048);                                                           \
        GQ_NWAIT(8, X##a2l, X##a2h);                                                                  \
        GQ_NMF(c20, X##a2l, X##a2h, X##b0l, X##b0h);                                                  \
        GQ_NRD(la7l, AC, 7, (OFFC) + 0);                                                              \
        GQ_NMF(c21, X##a2l, X##a2h, X##b1l, X##b1h);                                                  \
        GQ_NRD(la7h, AC, 7, (OFFC) + 2048);                                                           \
        GQ_NMF(c22, X##a2l, X##a2h, X##b2l, X##b2h);                                                  \
        GQ_NBAR();                                                                                    \
        GQ_NMF(c23, X##a2l, X##a2h, X##b3l, X##b3h);                                                  \
        L0;                                                                                           \
        GQ_NWAIT(8, la3l, la3h);                                                                      \
        GQ_NMF(c30, la3l, la3h, X##b0l, X##b0h);                                                      \
        GQ_NRD(Y##a0l, AN, 0, (OFFN) + 0);                                                            \
        GQ_NMF(c31, la3l, la3h, X##b1l, X##b1h);                                                      \
        L1;                                                                                           \
        GQ_NRD(Y##a0h, AN, 0, (OFFN) + 2048);                                                         \
        GQ_NMF(c32, la3l, la3h, X##b2l, X##b2h);                                                      \
        GQ_NRD(Y##b0l, BN, 0, (OFFN) + 0);                                                            \
        GQ_NMF(c33, la3l, la3h, X##b3l, X##b3h);                                                      \
        L2;                                                                                           \
        GQ_NRD(Y##b0h, BN, 0, (OFFN) + 2048);                                                         \
        GQ_NWAIT(10, la4l, la4h);                                                                     \
        GQ_NMF(c40, la4l, la4h, X##b0l, X##b0h);                                                      \
        GQ_NRD(Y##b1l, BN, 1, (OFFN) + 0);                                                            \
        GQ_NMF(c41, la4l, la4h, X##b1l, X##b1h);                                                      \
        L3;                                                                                           \
        GQ_NRD(Y##b1h, BN, 1, (OFFN) + 2048);                                                         \
        GQ_NMF(c42, la4l, la4h, X##b2l, X##b2h);                                                      \
        GQ_NRD(Y##b2l, BN, 2, (OFFN) + 0);                                                            \
        GQ_NMF(c43, la4l, la4h, X##b3l, X##b3h);                                                      \
        GQ_NRD(Y##b2h, BN, 2, (OFFN) + 2048);                                                         \
        GQ_NWAIT(12, la5l, la5h);                                                                     \
        GQ_NMF(c50, la5l, la5h, X##b0l, X##b0h);                                                      \
        GQ_NRD(Y##b3l, BN, 3, (OFFN) + 0);                                                            \
        GQ_NMF(c51, la5l, la5h, X##b1l, X##b1h);                                                      \
        GQ_NRD(Y##b3h, BN, 3, (OFFN) + 2048);                                                         \
        GQ_NMF(c52, la5l, la5h, X##b2l, X##b2h);                                                      \
        GQ_NRD(Y##a1l, AN, 1, (OFFN) + 0);                                                            \
        GQ_NMF(c53, la5l, la5h, X##b3l, X##b3h);                                                      \
        GQ_NWAIT(13, la6l, la6h);                                                                     \
        GQ_NRD(Y##a1h, AN, 1, (OFFN) + 2048);                                                         \
        GQ_NMF(c60, la6l, la6h, X##b0l, X##b0h);                                                      \
        GQ_NRD(Y##a2l, AN, 2, (OFFN) + 0);                                                            \
        GQ_NMF(c61, la6l, la6h, X##b1l, X##b1h);                                                      \
        GQ_NWAIT(13, la7l, la7h);                                                                     \
        GQ_NRD(Y##a2h, AN, 2, (OFFN) + 2048);                                                         \
        GQ_NMF(c62, la6l, la6h, X##b2l, X##b2h);                                                      \
        GQ_NMF(c63, la6l, la6h, X##b3l, X##b3h);                                                      \
        GQ_NMF(c70, la7l, la7h, X##b0l, X##b0h);                                                      \
        GQ_NMF(c71, la7l, la7h, X##b1l, X##b1h);                                                      \
        GQ_NMF(c72, la7l, la7h, X##b2l, X##b2h);                                                      \
        GQ_NMF(c73, la7l, la7h, X##b3l, X##b3h);                                                      \
    } while (0)
#define GQ_NINTERVAL(X, Y, AC, OFFC, AN, BN, OFFN, SLOT3)                                             \
    GQ_NSTEP(X, Y, AC, OFFC, AN, BN, OFFN, GQ_NDL(voff0, SLOT3, 0), GQ_NDL(voff1, SLOT3, 1),          \
             GQ_NDL(voff2, SLOT3, 2), GQ_NDL(voff3, SLOT3, 3));                                       \
    GQ_NADV();

    for (int h = 0; h < 3; ++h) {
        GQ_NDL_(voff0, h, 0); GQ_NDL_(voff1, h, 1); GQ_NDL_(voff2, h, 2); GQ_NDL_(voff3, h, 3);
        GQ_NADV();
    }
    asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    GQ_NRD(pa0l, bAlo, 0, 0); GQ_NRD(pa0h, bAlo, 0, 2048);
    GQ_NRD(pb0l, bBlo, 0, 0); GQ_NRD(pb0h, bBlo, 0, 2048);
    GQ_NRD(pb1l, bBlo, 1, 0); GQ_NRD(pb1h, bBlo, 1, 2048);
    GQ_NRD(pb2l, bBlo, 2, 0); GQ_NRD(pb2h, bBlo, 2, 2048);
    GQ_NRD(pb3l, bBlo, 3, 0); GQ_NRD(pb3h, bBlo, 3, 2048);
    GQ_NRD(pa1l, bAlo, 1, 0); GQ_NRD(pa1h, bAlo, 1, 2048);
    GQ_NRD(pa2l, bAlo, 2, 0); GQ_NRD(pa2h, bAlo, 2, 2048);
    for (int n = 0; n < nhs; n += 4) {  // nhs % 4 == 0; interval n multiplies slot n & 3
        GQ_NINTERVAL(p, q, bAlo, 0, bAlo, bBlo, 32768, 3)
        GQ_NINTERVAL(q, p, bAlo, 32768, bAhi, bBhi, 0, 0)
        GQ_NINTERVAL(p, q, bAhi, 0, bAhi, bBhi, 32768, 1)
        GQ_NINTERVAL(q, p, bAhi, 32768, bAlo, bBlo, 0, 2)
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#undef GQ_NDL
#undef GQ_NDL_
#undef GQ_NADV
#undef GQ_NRD
#undef GQ_NWAIT
#undef GQ_NMF
#undef GQ_NBAR
#undef GQ_NSTEP
#undef GQ_NINTERVAL
    float* __restrict__ H = P.H;
    const float beta = P.beta, alpha = P.alpha;
    const int lr = lane & 15, lk = lane >> 4;
    const int64_t i0 = ti * BT + wm * 128 + 4 * lk, j0 = tj * BT + wn * 64 + lr;
    float* __restrict__ Pp = (aux != 0xffffffffu) ? grp.partial + (size_t)(aux >> 12) * (BT * BT) : nullptr;
    const int rl0 = wm * 128 + 4 * lk, cl0 = wn * 64 + lr;  // position inside the tile
#define GQ_NSTORE(c, i, j)                                                                            \
    do {                                                                                              \
        if (Pp) { /* K-split unit: raw sums, combined in fixed order by syrk_reduce_kernel */          \
            float* q_ = Pp + (rl0 + (i) * 16) * BT + cl0 + (j) * 16;                                  \
            q_[0] = c[0]; q_[BT] = c[1]; q_[2 * BT] = c[2]; q_[3 * BT] = c[3];                        \
            break;                                                                                    \
        }                                                                                             \
        const int64_t col = j0 + (j) * 16, row = i0 + (i) * 16;                                       \
        float4 h;                                                                                     \
        h.x = beta * H[(row + 0) * C + col] + alpha * c[0];                                           \
        h.y = beta * H[(row + 1) * C + col] + alpha * c[1];                                           \
        h.z = beta * H[(row + 2) * C + col] + alpha * c[2];                                           \
        h.w = beta * H[(row + 3) * C + col] + alpha * c[3];                                           \
        H[(row + 0) * C + col] = h.x; H[(row + 1) * C + col] = h.y;                                   \
        H[(row + 2) * C + col] = h.z; H[(row + 3) * C + col] = h.w;                                   \
        if (ti != tj) *reinterpret_cast<float4*>(H + col * C + row) = h;                              \
    } while (0)
    GQ_NSTORE(c00, 0, 0);
    GQ_NSTORE(c01, 0, 1);
    GQ_NSTORE(c02, 0, 2);
    GQ_NSTORE(c03, 0, 3);
    GQ_NSTORE(c10, 1, 0);
    GQ_NSTORE(c11, 1, 1);
    GQ_NSTORE(c12, 1, 2);
    GQ_NSTORE(c13, 1, 3);
    GQ_NSTORE(c20, 2, 0);
    GQ_NSTORE(c21, 2, 1);
    GQ_NSTORE(c22, 2, 2);
    GQ_NSTORE(c23, 2, 3);
    GQ_NSTORE(c30, 3, 0);
    GQ_NSTORE(c31, 3, 1);
    GQ_NSTORE(c32, 3, 2);
    GQ_NSTORE(c33, 3, 3);
    GQ_NSTORE(c40, 4, 0);
    GQ_NSTORE(c41, 4, 1);
    GQ_NSTORE(c42, 4, 2);
    GQ_NSTORE(c43, 4, 3);
    GQ_NSTORE(c50, 5, 0);
    GQ_NSTORE(c51, 5, 1);
    GQ_NSTORE(c52, 5, 2);
    GQ_NSTORE(c53, 5, 3);
    GQ_NSTORE(c60, 6, 0);
    GQ_NSTORE(c61, 6, 1);
    GQ_NSTORE(c62, 6, 2);
    GQ_NSTORE(c63, 6, 3);
    GQ_NSTORE(c70, 7, 0);
    GQ_NSTORE(c71, 7, 1);
    GQ_NSTORE(c72, 7, 2);
    GQ_NSTORE(c73, 7, 3);
#undef GQ_NSTORE
    }  // tile loop
}

// -------------- 16-bit SYRK, 256x256 tiles, FOUR waves with 128 x 128 wave tiles (option syrk_w4 = 1)
// The vendor library's shape for this product (hipBLASLt MT256x256x32, MIWT8_8: DESIGN.md K1) on this kernel's ring:
// same 4-slot LDS ring in the natural activation layout, same DMA pieces, same tile table / K-split / rendezvous and the
// same k order per accumulator as syrk16_256n_kernel -- results are bit-identical -- but 2 x 2 waves, one per SIMD, each
// with 8 x 8 accumulators of 16x16 in 256 AGPRs: 16 fragments per 64 MFMAs instead of 12 per 32, i.e. a third fewer LDS
// read bytes per flop.  Every wave brings token rows 8 w .. 8 w + 7 of BOTH operands (8 DMA pieces per k32 step).
// Stream of one k32 step (64 MFMAs, row-major over the A fragments; B fragments double-buffered, A fragments single-
// buffered: a_i of the next step is read right after row i of this step has issued):
//   MFMA 0,1: reads of a7 (this step's, ring slot n)      MFMA 2: s_waitcnt vmcnt(8) + s_barrier (half-stage n+1 landed,
//   everybody is through with slot n-1)                    row r: a_(r-1) and b_r of half-stage n+1 behind MFMAs 8r+1,2,4,5
//   DMA pieces of half-stage n+3 behind MFMAs 6, 14, .., 62.
// The order of a step's 32 fragment reads and 8 DMA pieces, as a table the waits are derived from.  Read codes:
// 0..15 = A fragment code >> 1, half code & 1 (fragment 7: of THIS step's ring slot, the others: of the next one);
// 16..31 = B fragment (code - 16) >> 1 of the next ring slot.
// (Tried and removed, r04: on DIAGONAL tiles the lower-left wave idle -- its quadrant is the transpose of the upper-right one, a
// quarter of those tiles' MFMA energy, ~1 % of a launch -- as a second, MFMA-free loop for that wave: the extra path pushed the
// register allocation over the edge, scratch reloads inside the hand-counted loop, half the speed.)
// (Measured on this table and removed, profiles/r04_syrk_w4_ab.txt: the whole B set read in rows 0-3, a step without its
// barrier, a step that never waits for its DMA -- all within 1 % of this order: no wait left that matters, as in the
// eight-wave kernel.)
struct W4Sched {
    signed char rd[64][4];  // reads issued behind MFMA m (-1: none)
    signed char dma[64];    // DMA piece issued behind MFMA m: 0-3 A pieces, 4-7 B pieces (-1: none)
    int bar;                // the step's barrier stands behind this MFMA
};
constexpr void w4_put(W4Sched& S, int m, int code) {
    for (int q = 0; q < 4; ++q)
        if (S.rd[m][q] < 0) {
            S.rd[m][q] = (signed char)code;
            return;
        }
}
constexpr W4Sched w4_sched() {
    W4Sched S{};
    for (int m = 0; m < 64; ++m) {
        S.dma[m] = -1;
        for (int q = 0; q < 4; ++q) S.rd[m][q] = -1;
    }
    S.bar = 2;
    w4_put(S, 0, 14);
    w4_put(S, 1, 15);
    for (int r = 1; r < 8; ++r) {  // a_(r-1) of the next step once row r-1 has issued
        w4_put(S, 8 * r + 1, 2 * (r - 1));
        w4_put(S, 8 * r + 2, 2 * (r - 1) + 1);
    }
    for (int r = 0; r < 8; ++r) {  // b_r of the next step in row r
        w4_put(S, 8 * r + 4, 16 + 2 * r);
        w4_put(S, 8 * r + 5, 16 + 2 * r + 1);
    }
    for (int r = 0; r < 8; ++r) S.dma[8 * r + 6] = (signed char)r;
    return S;
}
// position of read `code` in the step's issue order, reads issued before MFMA m
constexpr int w4_pos(const W4Sched& S, int code) {
    int n = 0;
    for (int m = 0; m < 64; ++m)
        for (int q = 0; q < 4; ++q) {
            if (S.rd[m][q] == code) return n;
            if (S.rd[m][q] >= 0) ++n;
        }
    return -1;
}
constexpr int w4_before(const W4Sched& S, int m) {
    int n = 0;
    for (int k = 0; k < m; ++k)
        for (int q = 0; q < 4; ++q)
            if (S.rd[k][q] >= 0) ++n;
    return n;
}
// lgkmcnt that guarantees read `code` (issued in the previous step, or in this one: `same`) has returned before MFMA m:
// the number of reads issued after it (LDS returns in order), at most 15
constexpr int w4_wait(const W4Sched& S, int code, int m, bool same) {
    const int n = same ? w4_before(S, m) - w4_pos(S, code) - 1 : (32 - w4_pos(S, code) - 1) + w4_before(S, m);
    return n < 15 ? n : 15;
}

#define GQ_WDL(vo, base, ldsa)                                                                        \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(vo), "s"(base), "s"(ldsa) : "memory")
#define GQ_WRD(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define GQ_WWAIT(N, x, y) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x), "+v"(y) : "n"(N) : "memory")

// registers of one wave of syrk16_256w_kernel (a local object whose members all live in registers: every index below
// is a template constant)
template <bool BF16>
struct W4 {
    f32x4 c[8][8];                           // accumulators (AGPRs)
    u32x2 al[8], ah[8], bl[2][8], bh[2][8];  // A fragments (one set), B fragments (two sets); l / h = token rows +0..3 / +4..7
    unsigned adA[2][8], adB[2][8];           // fragment addresses: [64 KiB half of the ring][fragment]
    unsigned voff[4];                        // DMA source offsets of the wave's four pieces per operand
    unsigned ldsw;                           // LDS address of the wave's first piece in ring slot 0
    const char *gA, *gB;                     // DMA bases of the half-stage to fetch next (wave-uniform)
    static constexpr W4Sched S = w4_sched();

    __device__ __forceinline__ void mfma(f32x4& acc, const u32x2& xl, const u32x2& xh, const u32x2& yl, const u32x2& yh) {
        const u32x4 A_ = __builtin_shufflevector(xl, xh, 0, 1, 2, 3), B_ = __builtin_shufflevector(yl, yh, 0, 1, 2, 3);
        if constexpr (BF16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(A_), "v"(B_));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(A_), "v"(B_));
    }
    template <int SC, int CODE>
    __device__ __forceinline__ void read() {
        if constexpr (CODE >= 0) {
            constexpr int SN = (SC + 1) & 3, PN = (SC & 1) ^ 1, f = (CODE & 15) >> 1, hi = CODE & 1;
            if constexpr (CODE >= 16) {
                if constexpr (hi) GQ_WRD(bh[PN][f], adB[SN >> 1][f], (SN & 1) * 32768 + 2048);
                else GQ_WRD(bl[PN][f], adB[SN >> 1][f], (SN & 1) * 32768);
            } else {
                constexpr int SS = f == 7 ? SC : SN;
                if constexpr (hi) GQ_WRD(ah[f], adA[SS >> 1][f], (SS & 1) * 32768 + 2048);
                else GQ_WRD(al[f], adA[SS >> 1][f], (SS & 1) * 32768);
            }
        }
    }
    // MFMA M of the k32 step on ring slot SC, and what is issued behind it
    template <int SC, int M>
    __device__ __forceinline__ void slot() {
        constexpr int SD = (SC + 3) & 3, PC = SC & 1;
        constexpr int i = M >> 3, j = M & 7;
        if constexpr (M < 8) GQ_WWAIT(w4_wait(S, 16 + 2 * j + 1, M, false), bl[PC][j], bh[PC][j]);
        if constexpr (j == 0 && i < 7) GQ_WWAIT(w4_wait(S, 2 * i + 1, M, false), al[i], ah[i]);
        if constexpr (M == 56) GQ_WWAIT(w4_wait(S, 15, M, true), al[7], ah[7]);
        mfma(c[i][j], al[i], ah[i], bl[PC][j], bh[PC][j]);
        if constexpr (M == S.bar) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        read<SC, S.rd[M][0]>();
        read<SC, S.rd[M][1]>();
        read<SC, S.rd[M][2]>();
        read<SC, S.rd[M][3]>();
        if constexpr (S.dma[M] >= 0) {  // a DMA piece of half-stage n+3
            constexpr int u = S.dma[M];
            if constexpr (u < 4) GQ_WDL(voff[u], gA, ldsw + (unsigned)(SD * S_BUF_BYTES + u * 1024));
            else GQ_WDL(voff[u - 4], gB, ldsw + (unsigned)(SD * S_BUF_BYTES + 16384 + (u - 4) * 1024));
        }
    }
    template <int SC, int... M>
    __device__ __forceinline__ void step_(std::integer_sequence<int, M...>) {
        (slot<SC, M>(), ...);
    }
    template <int SC>
    __device__ __forceinline__ void step() {
        step_<SC>(std::make_integer_sequence<int, 64>{});
    }
    // what a previous step would have left: a0..a6 and the B set of ring slot 0
    template <int F>
    __device__ __forceinline__ void first_a() {
        GQ_WRD(al[F], adA[0][F], 0);
        GQ_WRD(ah[F], adA[0][F], 2048);
    }
    template <int F>
    __device__ __forceinline__ void first_b() {
        GQ_WRD(bl[0][F], adB[0][F], 0);
        GQ_WRD(bh[0][F], adB[0][F], 2048);
    }
    __device__ __forceinline__ void first_fragments() {
        first_a<0>(); first_a<1>(); first_a<2>(); first_a<3>(); first_a<4>(); first_a<5>(); first_a<6>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        first_b<0>(); first_b<1>(); first_b<2>(); first_b<3>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        first_b<4>(); first_b<5>(); first_b<6>(); first_b<7>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the step's waits count on a whole step of reads behind these
    }
    __device__ __forceinline__ void fetch(int h) {  // all eight pieces of a half-stage into ring slot h (prologue)
#pragma unroll
        for (int u = 0; u < 4; ++u) GQ_WDL(voff[u], gA, ldsw + (unsigned)(h * S_BUF_BYTES + u * 1024));
#pragma unroll
        for (int u = 0; u < 4; ++u) GQ_WDL(voff[u], gB, ldsw + (unsigned)(h * S_BUF_BYTES + 16384 + (u - 0) * 1024));
    }
};
#undef GQ_WDL
#undef GQ_WRD
#undef GQ_WWAIT

template <bool BF16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void syrk16_256w_kernel(const SyrkGroup grp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    int round = 0;
    for (int slot = (int)(blockIdx.x >> 3); slot < grp.per_xcd; slot += (int)(gridDim.x >> 3), ++round) {
    if (grp.bar && round > 0) {  // XCD-wide soft rendezvous between rounds, as in syrk16_256n_kernel
        if (tid == 0) {
            unsigned* b = grp.bar + (blockIdx.x & 7);
            __hip_atomic_fetch_add(b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)round * (gridDim.x >> 3);
            for (int spin = 0; spin < 64 && __hip_atomic_load(b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want; ++spin)
                __builtin_amdgcn_s_sleep(16);
        }
    }
    const uint32_t ent = (uint32_t)__builtin_amdgcn_readfirstlane((int)grp.table[(blockIdx.x & 7) * grp.per_xcd + slot]);
    if (ent == 0xffffffffu) {
        if (grp.bar) {
            if (grp.ck && tid == 0)
                __hip_atomic_fetch_add(grp.bar + 8 + (blockIdx.x & 7), (unsigned)grp.nck, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        break;
    }
    const uint32_t aux = grp.aux ? (uint32_t)__builtin_amdgcn_readfirstlane((int)grp.aux[(blockIdx.x & 7) * grp.per_xcd + slot])
                                 : 0xffffffffu;
    __syncthreads();
    const SyrkProblem& P = grp.p[ent >> 24];
    const int64_t ti = (ent >> 12) & 0xfff, tj = ent & 0xfff;
    const int64_t C = P.C;
    int64_t u0 = 0, u1 = P.Tp / (4 * SK);
    if (aux != 0xffffffffu) {
        const int64_t np = aux & 63, kp = (aux >> 6) & 63;
        u0 = kp * u1 / np;
        u1 = (kp + 1) * (P.Tp / (4 * SK)) / np;
    }
    const int nhs = (int)((u1 - u0) * 4);
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    // ---- DMA: wave w owns token rows 8 w .. 8 w + 7 of every half-stage, of both operands: four 1 KiB pieces (two rows
    // each) of the A image (channels 256 ti ..) and four of the B image (channels 256 tj ..)
    const int rb = 8 * wid;
    const int64_t hstride = 32 * C * 2;
    const int64_t cofA = ti * 512, cofB = tj * 512;
    const char* gsrc = reinterpret_cast<const char*>(P.Xt) + u0 * 4 * hstride;
    int hsps = P.hs_per_seg, seg_within = 0;
    const uint64_t* segp = P.segs;
    const char* segbase = nullptr;
    if (hsps) {
        const int h0 = (int)(u0 * 4);
        segp += h0 / hsps;
        seg_within = h0 % hsps;
        segbase = reinterpret_cast<const char*>(sload64_now(segp));
        gsrc = segbase + (int64_t)seg_within * hstride;
    }
    asm volatile("" : "+s"(hsps), "+s"(segp));
    W4<BF16> w;
    {
        const int hrow = lane >> 5, s16 = lane & 31;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r_ = rb + 2 * u + hrow;
            const int g_ = (r_ & 3) | (((r_ >> 3) & 1) << 2);
            w.voff[u] = (unsigned)(r_ * C * 2) + (unsigned)((((s16 >> 1) ^ g_) << 5) + ((s16 & 1) << 4));
        }
    }
    w.ldsw = lds0 + (unsigned)(rb * 512);
    int hnext = 0;
    w.gA = gsrc + cofA;
    w.gB = gsrc + cofB;
#define GQ_WADV()                                                                                     \
    do {                                                                                              \
        if (hnext + 1 < nhs) {                                                                        \
            ++hnext;                                                                                  \
            const char* g_;                                                                           \
            if (hsps) {                                                                               \
                if (++seg_within == hsps) {                                                           \
                    seg_within = 0;                                                                   \
                    ++segp;                                                                           \
                    segbase = reinterpret_cast<const char*>(sload64_now(segp));                       \
                }                                                                                     \
                g_ = segbase + (int64_t)seg_within * hstride;                                         \
            } else {                                                                                  \
                g_ = gsrc + (int64_t)hnext * hstride;                                                 \
            }                                                                                         \
            w.gA = g_ + cofA;                                                                         \
            w.gB = g_ + cofB;                                                                         \
        }                                                                                             \
    } while (0)
    // ---- fragment addresses: ring slot s = half s >> 1, offset (s & 1) * 32 KiB
    {
        const int kc = lane >> 4, q = (lane & 15) >> 2, ch = lane & 3;
        const int r = 8 * kc + q, g = q | ((kc & 1) << 2);
        const unsigned bA = lds0 + (unsigned)(r * 512 + (((wm * 8) ^ g) << 5) + ch * 8);
        const unsigned bB = lds0 + 16384u + (unsigned)(r * 512 + (((wn * 8) ^ g) << 5) + ch * 8);
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            w.adA[0][f] = bA ^ ((unsigned)f << 5);
            w.adB[0][f] = bB ^ ((unsigned)f << 5);
            w.adA[1][f] = w.adA[0][f] + 65536u;
            w.adB[1][f] = w.adB[0][f] + 65536u;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) w.c[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int h = 0; h < 3; ++h) {
        w.fetch(h);
        GQ_WADV();
    }
    asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
    w.first_fragments();
    const int ck = grp.bar ? grp.ck : 0;
    for (int n = 0; n < nhs; n += 4) {  // nhs % 4 == 0
        if (ck && n && (n & (ck - 1)) == 0 && tid == 0) {  // checkpoint n / ck of this round (SyrkGroup::ck)
            unsigned* b2 = grp.bar + 8 + (blockIdx.x & 7);
            __hip_atomic_fetch_add(b2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = ((unsigned)round * (unsigned)grp.nck + (unsigned)(n / ck)) * (gridDim.x >> 3);
            for (int spin = 0; spin < 32 && __hip_atomic_load(b2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want; ++spin)
                __builtin_amdgcn_s_sleep(4);
        }
        w.template step<0>(); GQ_WADV();
        w.template step<1>(); GQ_WADV();
        w.template step<2>(); GQ_WADV();
        w.template step<3>(); GQ_WADV();
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    if (ck && tid == 0) {  // what this unit did not pass of the round's nck checkpoints (K-split units are shorter)
        const int passed = (nhs - 1) / ck;
        if (grp.nck > passed)
            __hip_atomic_fetch_add(grp.bar + 8 + (blockIdx.x & 7), (unsigned)(grp.nck - passed), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#undef GQ_WADV
    float* __restrict__ H = P.H;
    const float beta = P.beta, alpha = P.alpha;
    const int lr = lane & 15, lk = lane >> 4;
    const int64_t i0 = ti * BT + wm * 128 + 4 * lk, j0 = tj * BT + wn * 128 + lr;
    float* __restrict__ Pp = (aux != 0xffffffffu) ? grp.partial + (size_t)(aux >> 12) * (BT * BT) : nullptr;
    const int rl0 = wm * 128 + 4 * lk, cl0 = wn * 128 + lr;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (Pp) {  // K-split unit: raw sums, combined in fixed order by syrk_reduce_kernel
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 v = w.c[i][j];
                float* q_ = Pp + (rl0 + i * 16) * BT + cl0 + j * 16;
                q_[0] = v[0]; q_[BT] = v[1]; q_[2 * BT] = v[2]; q_[3 * BT] = v[3];
            }
            continue;
        }
        // r05: the 32 loads of a row of accumulator tiles are in flight together (one exposed HBM round trip per row instead of one
        // per tile: the epilogue was 3-6 % of a launch on un-capped operands); same values, same operations per element
        const int64_t row = i0 + i * 16;
        float hv[8][4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t col = j0 + j * 16;
            hv[j][0] = H[(row + 0) * C + col]; hv[j][1] = H[(row + 1) * C + col];
            hv[j][2] = H[(row + 2) * C + col]; hv[j][3] = H[(row + 3) * C + col];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 v = w.c[i][j];
            const int64_t col = j0 + j * 16;
            float4 h;
            h.x = beta * hv[j][0] + alpha * v[0];
            h.y = beta * hv[j][1] + alpha * v[1];
            h.z = beta * hv[j][2] + alpha * v[2];
            h.w = beta * hv[j][3] + alpha * v[3];
            H[(row + 0) * C + col] = h.x; H[(row + 1) * C + col] = h.y;
            H[(row + 2) * C + col] = h.z; H[(row + 3) * C + col] = h.w;
            if (ti != tj) *reinterpret_cast<float4*>(H + col * C + row) = h;
        }
    }
    }  // tile loop
}

// K-split tiles: H tile = beta * H tile + alpha * (P_0 + P_1 + ... + P_{n-1}), partial sums added in index
// order (deterministic), mirrored below the diagonal.  list[2 i] = tile entry (problem << 24 | ti << 12 | tj),
// list[2 i + 1] = first slot << 8 | nparts.  Grid (tiles, 16): a block owns 16 rows of a tile.
__global__ __launch_bounds__(256) void syrk_reduce_kernel(const SyrkGroup grp, const uint32_t* __restrict__ list) {
    const uint32_t ent = list[2 * blockIdx.x], w = list[2 * blockIdx.x + 1];
    const SyrkProblem& P = grp.p[ent >> 24];
    const int64_t ti = (ent >> 12) & 0xfff, tj = ent & 0xfff, C = P.C;
    const int np = (int)(w & 0xff);
    const float* __restrict__ part = grp.partial + (size_t)(w >> 8) * (BT * BT);
    float* __restrict__ H = P.H;
    const float beta = P.beta, alpha = P.alpha;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = (int)threadIdx.x + q * 256, r = (int)blockIdx.y * 16 + idx / 64, c4 = (idx % 64) * 4;
        float4 a = *reinterpret_cast<const float4*>(part + r * BT + c4);
        for (int k = 1; k < np; ++k) {
            const float4 b = *reinterpret_cast<const float4*>(part + (size_t)k * (BT * BT) + r * BT + c4);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        float* hp = H + (ti * BT + r) * C + tj * BT + c4;
        float4 h = *reinterpret_cast<const float4*>(hp);
        h.x = beta * h.x + alpha * a.x; h.y = beta * h.y + alpha * a.y;
        h.z = beta * h.z + alpha * a.z; h.w = beta * h.w + alpha * a.w;
        *reinterpret_cast<float4*>(hp) = h;
        if (ti != tj) {
            float* mp = H + (tj * BT + c4) * C + ti * BT + r;
            mp[0] = h.x; mp[C] = h.y; mp[2 * C] = h.z; mp[3 * C] = h.w;
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

// 16-bit inputs with C % 256 == 0 and T % 128 == 0 are read in place (syrk16_256n_kernel): only the tile table
// needs scratch.  Everything else goes through the re-laid-out operand image.
static inline bool syrk_in_place(int64_t T, int64_t C) {
    return (C % BT == 0) && (T % (2 * HK) == 0) && !opt(OPT_syrk_image) && !opt(OPT_syrk_128);
}

// K-split of the last (partial) round of tiles (syrk16_256n_kernel): partial-sum slots a problem may need.
// Only long token ranges are split (every part is at least 8 turns of the ring); GQ_SYRK_NOSPLIT disables it.
static inline size_t syrk_partial_slots(int64_t T, size_t ntile) {
    if (T < 8192 || opt(OPT_syrk_nosplit)) return 0;
    return 4 * ntile < 512 ? 4 * ntile : 512;
}

size_t h_accumulate_workspace_bytes(int64_t T, int64_t C) {
    const size_t nt = (size_t)(C / BT);
    const size_t ntile = nt * (nt + 1) / 2;
    // tile table + unit attributes + reduce list of the 256x256 kernels (K-split adds up to 512 units)
    const size_t table = (ntile + 512 + 320) * 4 * 2 + 512 * 8 + 256 + (size_t)(T / 128 + 2) * 8 + 64;  // + block addresses, counters
    if (syrk_in_place(T, C)) return table + syrk_partial_slots(T, ntile) * (size_t)BT * BT * 4 + 256;
    const int64_t Tp = (T + 2 * HK - 1) / (2 * HK) * (2 * HK);
    return (size_t)C * (size_t)Tp * 2 + 256 + table;
}

// one launch over the problems idx[0..m): kind 0 = 128x128 kernel, 1 = 256x256 ring kernel on the re-laid-out
// image, 2 = 256x256 ring kernel reading X in place
static int syrk16_launch(int kind, const int* idx, int m, float* const* H, const void* const* X, const int64_t* T,
                         const int64_t* C, const float* beta, const float* alpha, int x_dtype, unsigned char*& wp,
                         unsigned char* ws_end, hipStream_t st, const void* const* const* segs = nullptr,
                         const int64_t* nseg = nullptr) {
    const dim3 block(256);
    SyrkGroup grp;
    grp.n = m;
    int tiles = 0;
    for (int k = 0; k < m; ++k) {
        const int i = idx[k];
        const int64_t Tp = (T[i] + 2 * HK - 1) / (2 * HK) * (2 * HK);  // whole turns of the 4-slot ring
        const uint16_t* Xt = reinterpret_cast<const uint16_t*>(X[i]);
        if (kind != 2) {
            uint16_t* img = reinterpret_cast<uint16_t*>(wp);
            wp += ((size_t)C[i] * Tp * 2 + 255) & ~(size_t)255;
            if (wp > ws_end) GQ_FAIL(GQ_E_WORKSPACE, "gq_h_accumulate: workspace too small for the operand image");
            ProfScope ps(PT_TRANSPOSE, st);
            dim3 tg((unsigned)(Tp / HK), (unsigned)(C[i] / HT));
            hipLaunchKernelGGL(transpose16_kernel, tg, block, 0, st, (const uint16_t*)X[i], T[i], C[i], img, Tp / HK,
                               kind == 1 ? 1 : 0);
            GQ_LAUNCH_CHECK();
            Xt = img;
        }
        const int nt = (int)(C[i] / (kind ? BT : HT));
        grp.p[k] = SyrkProblem{H[i], Xt, C[i], Tp, beta[i], alpha[i], tiles, nt, nullptr, 0};
        if (!kind) {
            const int ns = (nt + 7) / 8;  // 8x8 super-tiles per dimension
            tiles += ns * (ns + 1) / 2;
        }
    }
    grp.total_tiles = tiles;
    grp.table = nullptr;
    grp.per_xcd = 0;
    grp.aux = nullptr;
    grp.partial = nullptr;
    grp.bar = nullptr;
    grp.ck = 0;
    grp.nck = 0;
    int n_reduce = 0;
    const uint32_t* reduce_list = nullptr;
    std::vector<uint32_t> table;
    if (kind) {
        // Balanced schedule: the valid (ti <= tj) tiles of all problems, enumerated super-tile by super-tile
        // (8 x 4 tiles share 12 operand panels), are cut into groups of 32 CONSECUTIVE tiles -- one group =
        // what the 32 CUs of an XCD run at the same time out of one L2 -- and the groups are dealt
        // round-robin to the 8 XCDs, so every XCD gets the same number of tiles (+-32) and no workgroup
        // exits early.  (An arithmetic id -> super-tile map leaves XCDs up to 30 % apart: diagonal
        // super-tiles are half empty.)
        std::vector<uint32_t> all;
        // (option syrk_gw: the super-tile's width in tiles, 32 / gw rows.  r06 A/B of the fabric traffic on random operands at
        // C = 14336, profiles/r06_syrk_traffic_ab.txt: 4 x 8 (r02-r05) 29.7 GB of L2-miss reads per launch, 1312 TFLOP/s; 2 x 16
        // 41.5 GB, 1283; 1 x 32 51.3 GB, 1269; 8 x 4 (default since r06) 26.5 GB, 1332 -- and 89.3 against 90.2 ms per bench
        // step in four alternating pairs.  Which tiles fall into the K-split round depends on the shape, so H moves within
        // K1's tolerance class.)
        const int gw = (int)opt(OPT_syrk_gw), gh = 32 / gw;
        for (int k = 0; k < m; ++k) {
            const int nt = grp.p[k].nt, nsc = (nt + gw - 1) / gw, nsr = (nt + gh - 1) / gh;
            for (int sI = 0; sI < nsr; ++sI)
                for (int sJ = (sI * gh) / gw; sJ < nsc; ++sJ)
                    for (int slot = 0; slot < 32; ++slot) {
                        const int ti = sI * gh + slot / gw, tj = sJ * gw + slot % gw;
                        if (ti < nt && tj < nt && ti <= tj) all.push_back((uint32_t)k << 24 | (uint32_t)ti << 12 | (uint32_t)tj);
                    }
        }
        // K-split of the last, partial round (in-place kernel): 1596 tiles of a 14336-wide Hessian are 6.23 rounds
        // of 256 CUs, and the 7th round would run 60 tiles on 256 CUs for a whole tile time.  The R = N mod 256
        // tiles left after the full rounds are cut into s token ranges each, s minimising ceil(R s / 256) / s
        // (14336: s = 4, +0.25 instead of +1 round; 3 x 4096: R = 152, s = 3, +0.67 instead of +1); every unit
        // stores raw fp32 sums and syrk_reduce_kernel adds them in fixed order (deterministic).
        const size_t N = all.size(), full = (N / 256) * 256, R = N - full;
        int sp = 1;
        size_t cap_units = 0;
        bool can_split = kind == 2 && R > 0;
        for (int k = 0; k < m && can_split; ++k) {
            const size_t slots = syrk_partial_slots(grp.p[k].Tp, (size_t)grp.p[k].nt * (grp.p[k].nt + 1) / 2);
            if (slots == 0) can_split = false;
            cap_units += slots;
        }
        if (cap_units > 512) cap_units = 512;  // (measured, r04: 1024 -- s = 5 instead of 3 for the three 4096-wide inputs of a dense block, 1.6 rounds instead of 1.667 -- is 1-2 % SLOWER: partial-sum traffic and a longer reduce)
        if (can_split) {
            double best = 1.0;
            for (int c = 2; c <= 8 && R * c <= cap_units; ++c) {
                const double cost = (double)((R * c + 255) / 256) / c;
                if (cost < best - 1e-9) { best = cost; sp = c; }
            }
        }
        std::vector<uint32_t> unit, uaux, rlist;
        for (size_t t = 0; t < N; ++t) {
            if (t < full || sp == 1) {
                unit.push_back(all[t]);
                uaux.push_back(0xffffffffu);
            } else {
                const uint32_t first = (uint32_t)((t - full) * sp);
                rlist.push_back(all[t]);
                rlist.push_back(first << 8 | (uint32_t)sp);
                for (int k = 0; k < sp; ++k) {
                    unit.push_back(all[t]);
                    uaux.push_back((first + k) << 12 | (uint32_t)k << 6 | (uint32_t)sp);
                }
            }
        }
        const size_t ngroups = (unit.size() + 31) / 32;
        const int per_xcd = (int)((ngroups + 7) / 8) * 32;
        const size_t tl = (size_t)8 * per_xcd;
        table.assign(2 * tl + rlist.size(), 0xffffffffu);  // [table | aux | reduce list]
        for (size_t t = 0; t < unit.size(); ++t) {
            const size_t gi = t >> 5, pos = (gi & 7) * per_xcd + (gi >> 3) * 32 + (t & 31);
            table[pos] = unit[t];
            table[tl + pos] = uaux[t];
        }
        for (size_t t = 0; t < rlist.size(); ++t) table[2 * tl + t] = rlist[t];
        wp = reinterpret_cast<unsigned char*>(((uintptr_t)wp + 255) & ~(uintptr_t)255);
        // rendezvous counters of the persistent launch (zeroed by this upload), then the block address lists of the
        // problems whose X arrives in separate blocks, behind the tile table
        // default on: L2-miss reads of a 14336-wide launch 49.6 -> 37.5 GB (rocprofv3 FETCH_SIZE), +0.8 % speed
        const bool persist = opt(OPT_syrk_persist) != 0;
        size_t bar_at = 0;
        if (kind == 2 && persist && per_xcd > 32) {
            bar_at = table.size();
            for (int x = 0; x < 16; ++x) table.push_back(0u);  // [0..7] rounds, [8..15] checkpoints inside a tile
            // checkpoints: only when every problem of the launch walks the same number of half-stages (a dense block's inputs do;
            // MoE experts with their own token counts do not)
            int64_t ck = opt(OPT_syrk_ck);
            bool same = (ck & (ck - 1)) == 0 && ck >= 16 && opt(OPT_syrk_w4) != 0;
            for (int k = 1; k < m && same; ++k) same = grp.p[k].Tp == grp.p[0].Tp;
            if (same && grp.p[0].Tp / 32 > ck) {
                grp.ck = (int)ck;
                grp.nck = (int)((grp.p[0].Tp / 32 - 1) / ck);
            }
        }
        if (table.size() & 1) table.push_back(0xffffffffu);
        for (int k = 0; k < m && segs; ++k) {
            const int i = idx[k];
            if (!segs[i] || nseg[i] <= 1) continue;
            grp.p[k].segs = reinterpret_cast<const uint64_t*>(wp + table.size() * 4);
            grp.p[k].hs_per_seg = (int)(T[i] / nseg[i] / 32);
            for (int64_t b = 0; b < nseg[i]; ++b) {
                const uint64_t a = (uint64_t)(uintptr_t)segs[i][b];
                table.push_back((uint32_t)a);
                table.push_back((uint32_t)(a >> 32));
            }
        }
        if (wp + table.size() * 4 > ws_end) GQ_FAIL(GQ_E_WORKSPACE, "gq_h_accumulate: workspace too small for the tile table");
        GQ_HIP(hipMemcpyAsync(wp, table.data(), table.size() * 4, hipMemcpyHostToDevice, st));  // pageable: staged before return
        grp.table = reinterpret_cast<const uint32_t*>(wp);
        if (bar_at) grp.bar = reinterpret_cast<unsigned*>(wp) + bar_at;
        grp.per_xcd = per_xcd;
        grp.aux = sp > 1 ? grp.table + tl : nullptr;
        wp += table.size() * 4;
        if (sp > 1) {
            wp = reinterpret_cast<unsigned char*>(((uintptr_t)wp + 255) & ~(uintptr_t)255);
            grp.partial = reinterpret_cast<float*>(wp);
            wp += R * sp * (size_t)BT * BT * 4;
            if (wp > ws_end) GQ_FAIL(GQ_E_WORKSPACE, "gq_h_accumulate: workspace too small for the K-split partial sums");
            n_reduce = (int)R;
            reduce_list = grp.table + 2 * tl;
        }
    }
    ProfScope ps(PT_SYRK, st);
    const bool bf = x_dtype == GQ_BF16;
    if (kind == 2) {
        // one tile per workgroup, or (grp.bar) one workgroup per CU walking its XCD's list in rounds
        // (option syrk_wgs, read per call: fewer resident workgroups for a fold that runs NEXT TO a latency-bound chain --
        // the block schedule's postponed folds -- so that the chain's kernels always find free CUs)
        int wgs = 256;
        if (const int64_t e = opt(OPT_syrk_wgs)) wgs = (e >= 8 && e <= 256) ? (int)(e & ~7) : 256;
        const dim3 grid((unsigned)(grp.bar ? wgs : 8 * grp.per_xcd)), blk(512);
        if (opt(OPT_syrk_w4)) {  // four waves, 128 x 128 wave tiles (default; bit-identical to the eight-wave form)
            if (bf) hipLaunchKernelGGL(syrk16_256w_kernel<true>, grid, dim3(256), S_LDS_BYTES, st, grp);
            else hipLaunchKernelGGL(syrk16_256w_kernel<false>, grid, dim3(256), S_LDS_BYTES, st, grp);
        } else if (bf) hipLaunchKernelGGL(syrk16_256n_kernel<true>, grid, blk, S_LDS_BYTES, st, grp);
        else hipLaunchKernelGGL(syrk16_256n_kernel<false>, grid, blk, S_LDS_BYTES, st, grp);
        if (n_reduce > 0) {
            GQ_LAUNCH_CHECK();
            hipLaunchKernelGGL(syrk_reduce_kernel, dim3((unsigned)n_reduce, 16), dim3(256), 0, st, grp, reduce_list);
        }
    } else if (kind == 1) {
        const dim3 grid((unsigned)(8 * grp.per_xcd)), blk(512);
        if (bf) hipLaunchKernelGGL(syrk16_256e_kernel<true>, grid, blk, S_LDS_BYTES, st, grp);
        else hipLaunchKernelGGL(syrk16_256e_kernel<false>, grid, blk, S_LDS_BYTES, st, grp);
    } else {
        const dim3 grid((unsigned)((tiles + 7) / 8 * 8 * 64));
        if (bf) hipLaunchKernelGGL(syrk16_kernel<true>, grid, block, 2 * H_STAGE_BYTES, st, grp);
        else hipLaunchKernelGGL(syrk16_kernel<false>, grid, block, 2 * H_STAGE_BYTES, st, grp);
    }
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

// ---- the one collective of the path (reference gptq.py:131-132) moves the upper triangle only ----
// H is exactly symmetric, so the all-reduce needs (nt (nt + 1) / 2) 128 x 128 tiles instead of nt^2: pack copies the
// tiles ti <= tj into a contiguous buffer (row-major over the triangle), unpack writes them back and mirrors them
// below the diagonal -- half the bytes on the xGMI links, and the result is symmetric bit for bit on every rank
// whatever order the ring reduces in.
__global__ __launch_bounds__(256) void h_pack_upper_kernel(const float* __restrict__ H, int64_t C, float* __restrict__ buf) {
    const int64_t nt = C / HT;
    int64_t ti, tj;
    tri_tile(blockIdx.x, nt, ti, tj);
    const float* src = H + ti * HT * C + tj * HT;
    float* dst = buf + (size_t)blockIdx.x * (HT * HT);
#pragma unroll 4
    for (int idx = threadIdx.x; idx < HT * HT / 4; idx += 256) {
        const int r = idx / (HT / 4), c4 = (idx % (HT / 4)) * 4;
        *reinterpret_cast<float4*>(dst + r * HT + c4) = *reinterpret_cast<const float4*>(src + r * C + c4);
    }
}
__global__ __launch_bounds__(256) void h_unpack_upper_kernel(const float* __restrict__ buf, int64_t C, float* __restrict__ H) {
    __shared__ float tile[HT][HT + 1];
    const int64_t nt = C / HT;
    int64_t ti, tj;
    tri_tile(blockIdx.x, nt, ti, tj);
    const float* src = buf + (size_t)blockIdx.x * (HT * HT);
    float* dst = H + ti * HT * C + tj * HT;
    for (int idx = threadIdx.x; idx < HT * HT / 4; idx += 256) {
        const int r = idx / (HT / 4), c4 = (idx % (HT / 4)) * 4;
        const float4 v = *reinterpret_cast<const float4*>(src + r * HT + c4);
        *reinterpret_cast<float4*>(dst + r * C + c4) = v;
        tile[r][c4] = v.x; tile[r][c4 + 1] = v.y; tile[r][c4 + 2] = v.z; tile[r][c4 + 3] = v.w;
    }
    if (ti == tj) return;  // a diagonal tile is symmetric itself (mirrored SYRK epilogue, elementwise reduction)
    __syncthreads();
    float* mir = H + tj * HT * C + ti * HT;
    for (int idx = threadIdx.x; idx < HT * HT / 4; idx += 256) {
        const int r = idx / (HT / 4), c4 = (idx % (HT / 4)) * 4;  // row r of the mirrored tile = column r of the tile
        *reinterpret_cast<float4*>(mir + r * C + c4) = make_float4(tile[c4][r], tile[c4 + 1][r], tile[c4 + 2][r], tile[c4 + 3][r]);
    }
}

// Staging copy of the hook side (GPTQ.update keeps the activations of up to 64 Ki tokens until they are folded into H
// by one long-K SYRK): a plain HBM-bound copy, 16-byte accesses, 4 loads in flight per thread.  A kernel instead
// of hipMemcpyAsync: the runtime's blit path costs 20-40 us per call and serialises on the stream, 512 calls per
// transformer block.
__global__ __launch_bounds__(256) void stage_rows_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, int64_t n16) {
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 * 4 + threadIdx.x; i < n16; i += stride) {
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + k * 256 < n16) v[k] = src[i + k * 256];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + k * 256 < n16) dst[i + k * 256] = v[k];
    }
}
__global__ __launch_bounds__(256) void stage_bytes_kernel(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}
// The staging copies of a whole fold in ONE launch: n source blocks (16-byte aligned, whole 16-byte units) land one
// behind the other at dst.  tab = [n source addresses | n + 1 prefix offsets in 16-byte units].  A ragged handle (an MoE
// expert: a data-dependent handful of tokens per calibration sample) otherwise pays one latency-bound launch per
// sample and Linear -- 2080 launches of ~26 us per Mixtral block, 160 GB/s.
__global__ __launch_bounds__(256) void stage_many_kernel(uint4* __restrict__ dst, const uint64_t* __restrict__ tab, int n, int64_t n16) {
    const uint64_t* pre = tab + n;
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (int64_t i0 = (int64_t)blockIdx.x * 256 * 4; i0 < n16; i0 += stride) {
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t i = i0 + k * 256 + threadIdx.x;
            if (i < n16) {
                int lo = 0, hi = n;  // pre[lo] <= i < pre[hi]
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if ((int64_t)pre[mid] <= i) lo = mid;
                    else hi = mid;
                }
                v[k] = reinterpret_cast<const uint4*>(tab[lo])[i - (int64_t)pre[lo]];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t i = i0 + k * 256 + threadIdx.x;
            if (i < n16) dst[i] = v[k];
        }
    }
}
int h_stage_many(void* dst, const void* const* srcs, const int64_t* nbytes, int n, void* ws, size_t ws_bytes, hipStream_t st) {
    if (n <= 0) return GQ_OK;
    if (!dst || !srcs || !nbytes || !ws) GQ_FAIL(GQ_E_NULL, "gq_h_stage_many: null pointer");
    if (ws_bytes < (size_t)(2 * n + 1) * 8 + 256) GQ_FAIL(GQ_E_WORKSPACE, "gq_h_stage_many: workspace %zu < %zu bytes", ws_bytes, (size_t)(2 * n + 1) * 8 + 256);
    std::vector<uint64_t> tab((size_t)2 * n + 1);
    uint64_t off = 0;
    if ((uintptr_t)dst % 16) GQ_FAIL(GQ_E_UNSUPPORTED, "gq_h_stage_many: dst is not 16-byte aligned");
    for (int k = 0; k < n; ++k) {
        if (nbytes[k] <= 0 || nbytes[k] % 16 || (uintptr_t)srcs[k] % 16)
            GQ_FAIL(GQ_E_UNSUPPORTED, "gq_h_stage_many: block %d (%ld bytes at %p) is not made of aligned 16-byte units", k, (long)nbytes[k], srcs[k]);
        tab[k] = (uint64_t)(uintptr_t)srcs[k];
        tab[n + k] = off;
        off += (uint64_t)nbytes[k] / 16;
    }
    tab[2 * n] = off;
    uint64_t* dtab = reinterpret_cast<uint64_t*>(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    GQ_HIP(hipMemcpyAsync(dtab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, st));  // pageable: staged before return
    const int64_t n16 = (int64_t)off;
    int64_t blocks = (n16 + 1023) / 1024;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(stage_many_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<uint4*>(dst), dtab, n, n16);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

int h_stage(void* dst, const void* src, int64_t nbytes, hipStream_t st, int max_wgs) {
    const int64_t cap = max_wgs > 0 ? max_wgs : 4096;
    if (nbytes < 0) GQ_FAIL(GQ_E_BAD_SHAPE, "gq_h_stage: nbytes=%ld", (long)nbytes);
    if (nbytes == 0) return GQ_OK;
    if (!dst || !src) GQ_FAIL(GQ_E_NULL, "gq_h_stage: null pointer");
    if ((((uintptr_t)dst | (uintptr_t)src | (uintptr_t)nbytes) & 15) == 0) {
        const int64_t n16 = nbytes / 16;
        const int64_t wgs = (n16 + 1023) / 1024;
        hipLaunchKernelGGL(stage_rows_kernel, dim3((unsigned)(wgs < cap ? wgs : cap)), dim3(256), 0, st, (uint4*)dst,
                           (const uint4*)src, n16);
    } else {
        const int64_t wgs = (nbytes + 255) / 256;
        hipLaunchKernelGGL(stage_bytes_kernel, dim3((unsigned)(wgs < cap ? wgs : cap)), dim3(256), 0, st, (uint8_t*)dst,
                           (const uint8_t*)src, nbytes);
    }
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

int h_pack_upper(const float* H, int64_t C, float* buf, hipStream_t st) {
    if (!H || !buf) GQ_FAIL(GQ_E_NULL, "gq_h_pack_upper: null pointer");
    if (C <= 0 || (C % HT)) GQ_FAIL(GQ_E_BAD_SHAPE, "gq_h_pack_upper: C=%ld (C %% 128 != 0)", (long)C);
    const int64_t nt = C / HT;
    hipLaunchKernelGGL(h_pack_upper_kernel, dim3((unsigned)(nt * (nt + 1) / 2)), dim3(256), 0, st, H, C, buf);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}
int h_unpack_upper(const float* buf, int64_t C, float* H, hipStream_t st) {
    if (!H || !buf) GQ_FAIL(GQ_E_NULL, "gq_h_unpack_upper: null pointer");
    if (C <= 0 || (C % HT)) GQ_FAIL(GQ_E_BAD_SHAPE, "gq_h_unpack_upper: C=%ld (C %% 128 != 0)", (long)C);
    const int64_t nt = C / HT;
    hipLaunchKernelGGL(h_unpack_upper_kernel, dim3((unsigned)(nt * (nt + 1) / 2)), dim3(256), 0, st, buf, C, H);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

int h_accumulate_grouped(int n, float* const* H, const void* const* X, const int64_t* T, const int64_t* C,
                         const float* beta, const float* alpha, int x_dtype, void* ws, size_t ws_bytes,
                         hipStream_t st, const void* const* const* segs, const int64_t* nseg) {
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
        if (segs && segs[i] && nseg[i] > 1) {
            // separate blocks are only read in place: whole ring turns per block, 16-byte aligned rows
            if (T[i] % nseg[i] || (T[i] / nseg[i]) % (2 * HK) || !syrk_in_place(T[i], C[i]) || opt(OPT_syrk_128))
                GQ_FAIL(GQ_E_UNSUPPORTED, "gq_h_accumulate_segments: %ld blocks of %ld tokens, C=%ld: blocks must hold a "
                        "multiple of 128 tokens and C %% 256 == 0 (stage the rows into one buffer instead)",
                        (long)nseg[i], (long)(T[i] / nseg[i]), (long)C[i]);
            for (int64_t b = 0; b < nseg[i]; ++b)
                if (!segs[i][b] || ((uintptr_t)segs[i][b] & 15)) GQ_FAIL(GQ_E_NULL, "gq_h_accumulate_segments: block %ld of problem %d is null or not 16-byte aligned", (long)b, i);
        }
    }
    if (!ws || ws_bytes < need) GQ_FAIL(GQ_E_WORKSPACE, "gq_h_accumulate: workspace %zu < %zu bytes", ws_bytes, need);
    static std::atomic<bool> attr_set{false};  // guards an idempotent call: a race sets the same value twice
    if (!attr_set) {
        GQ_HIP(hipFuncSetAttribute((const void*)syrk16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * H_STAGE_BYTES));
        GQ_HIP(hipFuncSetAttribute((const void*)syrk16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * H_STAGE_BYTES));
        GQ_HIP(hipFuncSetAttribute((const void*)syrk16_256e_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES));
        GQ_HIP(hipFuncSetAttribute((const void*)syrk16_256e_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES));
        GQ_HIP(hipFuncSetAttribute((const void*)syrk16_256n_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES));
        GQ_HIP(hipFuncSetAttribute((const void*)syrk16_256n_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES));
        GQ_HIP(hipFuncSetAttribute((const void*)syrk16_256w_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES));
        GQ_HIP(hipFuncSetAttribute((const void*)syrk16_256w_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES));
        attr_set = true;
    }
    // Problems are sorted into at most three launches by the kernel they can take (usually all take the first):
    // in place (kind 2), 256x256 on the image (kind 1: T not a multiple of 128), 128x128 (kind 0: C % 256 != 0).
    int idx[3][H_MAX_GROUP], cnt[3] = {0, 0, 0};
    const bool force128 = opt(OPT_syrk_128) != 0;
    for (int i = 0; i < n; ++i) {
        const int kind = (force128 || C[i] % BT) ? 0 : (syrk_in_place(T[i], C[i]) ? 2 : 1);
        idx[kind][cnt[kind]++] = i;
    }
    unsigned char* wp = reinterpret_cast<unsigned char*>(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    unsigned char* ws_end = reinterpret_cast<unsigned char*>(ws) + ws_bytes;
    for (int kind = 2; kind >= 0; --kind)
        if (cnt[kind]) {
            const int rc = syrk16_launch(kind, idx[kind], cnt[kind], H, X, T, C, beta, alpha, x_dtype, wp, ws_end, st,
                                         kind == 2 ? segs : nullptr, nseg);
            if (rc) return rc;
        }
    return GQ_OK;
}

int h_accumulate(float* H, const void* X, int x_dtype, int64_t T, int64_t C, float beta, float alpha, void* ws,
                 size_t ws_bytes, hipStream_t st) {
    return h_accumulate_grouped(1, &H, &X, &T, &C, &beta, &alpha, x_dtype, ws, ws_bytes, st, nullptr, nullptr);
}

}  // namespace gq
