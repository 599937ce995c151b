// gq_gptq.hip -- K5 (per-column quantize + in-block error feedback), K6 (blocked
// trailing update) and the per-Linear orchestration of GPTQ.step.
//
// Reference: gptq.py:145-276.  Rows of W are independent given U, so K5 gives one
// LANE to a row (a wave = 64 rows) and walks the columns of a block in program
// order; U is wave-uniform and arrives through the scalar cache.  K6 is the
// GEMM W[:, c2:] -= Err[R,B] @ U[c1:c2, c2:] on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32 == a k-ordered fmaf chain, which is what the reference's
// sgemm computes per element -- verified bit-for-bit in tests/golden G6).
#include <atomic>
#include <mutex>
#include "gq_common.hpp"
#include "gq_gemm32.hpp"
#include <stdlib.h>

namespace gq {

int launch_scale_search(const float* x, int64_t rows, int64_t ld, int q_type, const gq_search_t* p,
                        uint16_t* d, int64_t d_stride, uint8_t* s, int64_t s_ld, uint16_t* dmin,
                        int64_t dmin_stride, uint8_t* m, int64_t m_ld, hipStream_t st, unsigned* panel = nullptr,
                        const int64_t* row_ends = nullptr, int nstack = 1);

// ---------------------------------------------------------------- K5 segment
// Processes columns [a, a+len) (len <= 128, a % 16 == 0, len % 16 == 0) of the
// current block for 64 rows per wave.
//   src   : working values of these columns (W itself when the block is a single
//           segment, else the block scratch), row stride ld_src, already offset to
//           column a
//   W     : receives the dequantized column (gptq.py:266), row stride C
//   Err   : receives err (gptq.py:268) at column (a - c1) + i, row stride ld_err
//   qparams d/dmin [R, C/256], s/m [R, C/G] already hold the super-group of `a`
constexpr int SEG = 128;
constexpr int SB = 16;  // register sub-block
constexpr int SEG_WAVES = 8;  // wave 0 walks the dependent chain, all waves share the rank-1 tile updates
constexpr int SEG_LDS_BYTES = (SEG * 64 + SEG * SEG + 2 * SB * 64) * 4 + SEG * 8;

// Correctly rounded fp32 division through an fp64 reciprocal: fl32(fl64(n * fl64(1 / d))) == fl32(n / d) for all
// finite fp32 n, d != 0.  The exact quotient of two 24-bit significands is either representable or at least 2^-49
// (relative) away from every fp32 rounding boundary (a boundary has 25 significant bits, so n - m d is a non-zero
// multiple of the unit of a 49-bit product), while the two fp64 roundings move the product by at most 2^-52: the
// final rounding sees the same side of every boundary as the exact quotient does.  Three VALU ops in the dependent
// chain (cvt, mul_f64, cvt) instead of the 11-instruction v_div_scale / v_rcp / fma / v_div_fixup sequence -- the
// column loop performs two divisions per column and is bound by exactly that chain.
__device__ __forceinline__ float div_rcp64(float n, double rd) { return (float)((double)n * rd); }

// UNI: the uniform grid of EvoPress' FastOBQ (evopress/src/quant_utils.py:23-29) instead of a K-quant:
//   q = clamp(round(w / max(scale, 1e-9) + zero), 0, maxq),  w_hat = scale * (q - zero)
// with scale / zero fp32 [R, C / G] (uscale / uzero; d, s, dmin, m unused).
template <bool PERM, bool UNI = false>
__global__ __launch_bounds__(SEG_WAVES * 64) void gptq_segment_kernel(
    float* W, int64_t C, const float* src, int64_t ld_src,  // may alias (single-segment blocks)
    const float* __restrict__ U, int64_t a, int len, int64_t R,
    const uint16_t* __restrict__ d, const uint8_t* __restrict__ s, const uint16_t* __restrict__ dmin,
    const uint8_t* __restrict__ m, int G, int is_signed, float qmin, float qmax,
    uint8_t* __restrict__ qweight, float* __restrict__ Err, int64_t ld_err, int64_t err_col0,
    const int32_t* __restrict__ perm, const float* __restrict__ uscale = nullptr,
    const float* __restrict__ uzero = nullptr,  // PERM (act_order, gptq.py:211-216): column j takes the parameters of
                                                // the group of its ORIGINAL column perm[j]
    const float* __restrict__ Unext = nullptr,  // != nullptr: U[a .. a+127][a+128 .. a+255]; epilogue below
    int npair = 1) {  // 2 (r05; needs Unext, src == W + a): this launch ALSO walks the partner block a+128 .. a+255 once its
                      // epilogue has brought this block's errors there -- every workgroup owns its 64 rows in both blocks, so
                      // nothing but the workgroup's own stores has to be visible: one launch per 256-column group instead of two
    // One workgroup = 64 rows (lane = row) x SEG_WAVES waves.  Wave 0 walks the columns (the dependent chain) one
    // 16-column sub-block ("tile") at a time; the rank-1 updates of the later tiles run UNDER the next chain:
    //   iteration t:  wave 0: chain(t) -> -err of tile t into ne[t & 1]
    //                 waves 1..7: apply ne[(t-1) & 1] (the errors of tile t-1) to the tiles t+1.. (deferred one tile)
    //                 barrier;  all 8 waves: apply ne[t & 1] to tile t+1 only, two columns each;  barrier
    // Tile u thus receives the errors of tiles 0, 1, .., u-2 (deferred, iterations 1..u-1) and then of tile u-1 (the
    // split step of iteration u-1), in this order, each as the same (mul, add) pairs in the same k order as the
    // reference's successive addr_ calls (gptq.py:267): results are unchanged, the update time leaves the critical
    // path (41 -> see DESIGN.md K5).
    extern __shared__ __attribute__((aligned(16))) float seg_smem[];
    float* wl = seg_smem;                   // wl[j*64 + lane]: working copy, column-major (32 KiB)
    float* Us = seg_smem + SEG * 64;        // Us[i*SEG + j] = U[a+i, a+j]: diagonal block (64 KiB), read back
                                            // with wave-uniform (broadcast) 16-byte LDS loads
    float* ne = Us + SEG * SEG;             // ne[(t & 1)*SB*64 + k*64 + lane]: -err of tile t, double-buffered (8 KiB)
    double* rdiag = reinterpret_cast<double*>(ne + 2 * SB * 64);  // rdiag[i] = 1 / U[a+i, a+i] in fp64 (1 KiB)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t row = (int64_t)blockIdx.x * 64 + lane;
    const bool live = row < R;
    const int64_t r = live ? row : 0;
    for (int half = 0; half < npair; ++half) {
    if (half) {  // the partner block: same rows, next 128 columns; the epilogue's stores of this workgroup are complete
        __syncthreads();
        a += SEG;
        src += SEG;
        err_col0 += SEG;
        Unext = nullptr;
    }
    // Two-step prologue: everything the FIRST tile's chain needs (U rows 0..15, the tile's 16 columns of W, its
    // reciprocal diagonal) is brought in by all waves, then the chain starts while waves 1..7 -- idle in
    // iteration 0 -- bring in the other 7/8 (gptq.py:225 w_blk = w[:, c1:c2].clone()).
    auto load_u_rows = [&](int i_lo, int i_hi, int t0, int nthr) {
        for (int idx = t0; idx < (i_hi - i_lo) * (SEG / 4); idx += nthr) {
            const int i = i_lo + idx / (SEG / 4), j4 = (idx % (SEG / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j4 < len) v = *reinterpret_cast<const float4*>(U + (a + i) * C + a + j4);
            *reinterpret_cast<float4*>(Us + i * SEG + j4) = v;
        }
    };
    auto load_w_cols = [&](int j_lo, int j_hi, int w0, int nw) {
        const float* sp = src + r * ld_src;
        for (int j = j_lo + w0 * 4; j < j_hi; j += 4 * nw) {
            float4 v = *reinterpret_cast<const float4*>(sp + j);
            wl[(j + 0) * 64 + lane] = v.x;
            wl[(j + 1) * 64 + lane] = v.y;
            wl[(j + 2) * 64 + lane] = v.z;
            wl[(j + 3) * 64 + lane] = v.w;
        }
    };
    const int first = len < SB ? len : SB;
    if (tid < first) rdiag[tid] = 1.0 / (double)U[(a + tid) * C + a + tid];
    load_u_rows(0, first, tid, SEG_WAVES * 64);
    load_w_cols(0, first, wid, SEG_WAVES);
    const int64_t nsg = C / 256, ng = C / G;
    // group parameters of every tile of the segment, fetched before the chain starts (they do not depend on it): a
    // tile lies inside one group (16 | G), ds = f32(d) * s, dm = f32(dmin) * m, and 1 / max(ds, 1e-9) in fp64
    constexpr int NT = SEG / SB;
    float dst[PERM ? 1 : NT], dmt[PERM ? 1 : NT];
    double rdt[PERM ? 1 : NT];
    if constexpr (!PERM) {
        if (wid == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int64_t col0 = a + t * SB < a + len ? a + t * SB : a;
                if constexpr (UNI) {
                    dst[t] = uscale[r * ng + col0 / G];
                    dmt[t] = uzero[r * ng + col0 / G];
                } else {
                    dst[t] = h2f(d[r * nsg + col0 / 256]) * ival(s[r * ng + col0 / G], is_signed);
                    dmt[t] = h2f(dmin[r * nsg + col0 / 256]) * ival(m[r * ng + col0 / G], is_signed);
                }
                rdt[t] = 1.0 / (double)(dst[t] < 1e-9f ? 1e-9f : dst[t]);  // quant_utils.py:37 clamp_min(eps)
            }
        }
    }
    __syncthreads();
    if (wid != 0) {  // under the first chain
        const int ht = tid - 64;
        if (ht >= first && ht < len) rdiag[ht] = 1.0 / (double)U[(a + ht) * C + a + ht];
        load_u_rows(first, len, ht, (SEG_WAVES - 1) * 64);
        load_w_cols(first, len, wid - 1, SEG_WAVES - 1);
    }

    const int ntile = len / SB;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t >= ntile) break;
        const int i0 = t * SB;
        float* net = ne + (t & 1) * (SB * 64);
        if (wid == 0) {
            const int64_t col0 = a + i0;
            float ds = 0.f, dm = 0.f, dsv[PERM ? SB : 1], dmv[PERM ? SB : 1];
            double rden = 0.0, rdenv[PERM ? SB : 1];
            if constexpr (PERM) {
#pragma unroll
                for (int k = 0; k < SB; ++k) {
                    const int64_t pc = perm[col0 + k];
                    dsv[k] = h2f(d[r * nsg + pc / 256]) * ival(s[r * ng + pc / G], is_signed);
                    dmv[k] = h2f(dmin[r * nsg + pc / 256]) * ival(m[r * ng + pc / G], is_signed);
                    rdenv[k] = 1.0 / (double)(dsv[k] < 1e-9f ? 1e-9f : dsv[k]);
                }
            } else {
                ds = dst[t];
                dm = dmt[t];
                rden = rdt[t];
            }
            float wr[SB], nerr[SB], wq[SB];
            uint32_t qpack[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < SB; ++k) wr[k] = wl[(i0 + k) * 64 + lane];
#pragma unroll
            for (int k = 0; k < SB; ++k) {
                const float* urow = Us + (i0 + k) * SEG + i0;  // wave-uniform LDS address (broadcast)
                if constexpr (PERM) { ds = dsv[k]; dm = dmv[k]; rden = rdenv[k]; }
                float q;
                if constexpr (UNI) {
                    q = clampf(rintf(div_rcp64(wr[k], rden) + dm), qmin, qmax);  // fast_obq.py:173, quant_utils.py:23-25
                    wq[k] = ds * (q - dm);                                       // :174, quant_utils.py:28-29
                } else {
                    q = clampf(rintf(div_rcp64(wr[k] + dm, rden)), qmin, qmax);  // gptq.py:247-254
                    wq[k] = dequantize1(q, ds, dm);                              // :255-261
                }
                const float err = div_rcp64(wr[k] - wq[k], rdiag[i0 + k]);                 // :264
                const uint8_t qb = is_signed ? (uint8_t)(int8_t)q : (uint8_t)q;
                qpack[k >> 2] |= (uint32_t)qb << (8 * (k & 3));
                // :267 addr_(err, U[i, i:], alpha=-1): self + (alpha*err)*u, two roundings
                const float nk = -err;
                nerr[k] = nk;
#pragma unroll
                for (int kk = k; kk < SB; ++kk) wr[kk] = wr[kk] + nk * urow[kk];
            }
            // published after the chain: an LDS store inside it would order every later LDS read of U behind it
            // (the compiler cannot tell that `ne` and `Us` never overlap)
#pragma unroll
            for (int k = 0; k < SB; ++k) net[k * 64 + lane] = nerr[k];
            if (live) {
                *reinterpret_cast<uint4*>(qweight + row * C + col0) =
                    make_uint4(qpack[0], qpack[1], qpack[2], qpack[3]);
                float* wp = W + row * C + col0;
                float* ep = Err + row * ld_err + err_col0 + i0;
#pragma unroll
                for (int k = 0; k < SB; k += 4) {
                    *reinterpret_cast<float4*>(wp + k) = make_float4(wq[k], wq[k + 1], wq[k + 2], wq[k + 3]);
                    *reinterpret_cast<float4*>(ep + k) =
                        make_float4(-nerr[k], -nerr[k + 1], -nerr[k + 2], -nerr[k + 3]);
                }
            }
        }
        // deferred: the errors of tile t-1 go into the tiles t+1.., 16 columns per helper wave and turn; each element
        // sees the same sequence of (mul, add) pairs, in the same order of i, as 16 successive addr_ calls
        // (packed v_pk_mul_f32 / v_pk_add_f32 were measured 35 % SLOWER here)
        if (wid != 0 && t > 0) {
            const float* nep = ne + ((t - 1) & 1) * (SB * 64);
            if (Unext != nullptr && wid == SEG_WAVES - 1) {
                // the epilogue wants all 128 (negated) errors of the block: tile t-1's go where its working columns were
                // (dead since its chain started), off the chain's wave
#pragma unroll
                for (int k = 0; k < SB; ++k) wl[(i0 - SB + k) * 64 + lane] = nep[k * 64 + lane];
            }
            int turn = 0;
            for (int j0 = i0 + SB; j0 < len; j0 += SB, ++turn) {
                if ((turn % (SEG_WAVES - 1)) + 1 != wid) continue;
                float wt[SB], nk[SB];
#pragma unroll
                for (int jj = 0; jj < SB; ++jj) wt[jj] = wl[(j0 + jj) * 64 + lane];
#pragma unroll
                for (int k = 0; k < SB; ++k) nk[k] = nep[k * 64 + lane];
#pragma unroll
                for (int k = 0; k < SB; ++k) {
                    const float* urow = Us + (i0 - SB + k) * SEG + j0;
#pragma unroll
                    for (int jj = 0; jj < SB; ++jj) wt[jj] = wt[jj] + nk[k] * urow[jj];
                }
#pragma unroll
                for (int jj = 0; jj < SB; ++jj) wl[(j0 + jj) * 64 + lane] = wt[jj];
            }
        }
        __syncthreads();
        // the next tile needs this tile's errors NOW: all eight waves, two of its columns each
        if (i0 + SB < len) {
            constexpr int CPW = SB / SEG_WAVES;
            const int j0 = i0 + SB + wid * CPW;
            float wt[CPW];
#pragma unroll
            for (int jj = 0; jj < CPW; ++jj) wt[jj] = wl[(j0 + jj) * 64 + lane];
#pragma unroll
            for (int k = 0; k < SB; ++k) {
                const float nk = net[k * 64 + lane];
#pragma unroll
                for (int jj = 0; jj < CPW; ++jj) wt[jj] = wt[jj] + nk * Us[(i0 + k) * SEG + j0 + jj];
            }
#pragma unroll
            for (int jj = 0; jj < CPW; ++jj) wl[(j0 + jj) * 64 + lane] = wt[jj];
        }
        __syncthreads();
    }
    // ---- epilogue (r03): this block's errors into the NEXT block's 128 columns (its partner in the 256-column
    // scale-search group), W[rows, a+128 ..] -= E[64 x 128] U[a .., a+128 ..], on the matrix cores: per element a
    // k-ordered fma chain from 0 and one subtraction, exactly what the separate rank-128 launch (gemm32_kernel, MODE 0)
    // computed -- here with the negated errors the kernel already keeps, sum' = -sum exactly, w + sum' = w - sum -- so
    // the launch, its gap and its W round trip disappear.  U's diagonal block in LDS is dead once the chains are through.
    if (Unext != nullptr) {
        for (int idx = tid; idx < SEG * (SEG / 4); idx += SEG_WAVES * 64) {
            const int i = idx / (SEG / 4), j4 = (idx % (SEG / 4)) * 4;
            *reinterpret_cast<float4*>(Us + i * SEG + j4) = *reinterpret_cast<const float4*>(Unext + (int64_t)i * C + j4);
        }
        __syncthreads();
        const int rb = (wid & 1) * 32, cb = (wid >> 1) * 32;  // 2 x 4 sub-tiles of 32 x 32, one per wave
        const int li = lane & 31, lk = lane >> 5;
        const float* nel = ne + ((ntile - 1) & 1) * (SB * 64);  // the last tile's errors never moved
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll 8
        for (int k0 = 0; k0 < SEG; k0 += 2) {
            const int k = k0 + lk;
            const float ev = (k < SEG - SB) ? wl[k * 64 + rb + li] : nel[(k - (SEG - SB)) * 64 + rb + li];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ev, Us[k * SEG + cb + li], acc, 0, 0, 0);
        }
        // all 16 loads in flight before the first store (written as `*p += acc` the compiler must keep every load behind
        // the previous store: 16 dependent HBM round trips, 12 us per block)
        const int64_t r0 = (int64_t)blockIdx.x * 64 + rb + 4 * lk;
        float* wp = W + r0 * C + a + SEG + cb + li;
        float wv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int ro = (e & 3) + 8 * (e >> 2);
            wv[e] = (r0 + ro < R) ? wp[(int64_t)ro * C] : 0.0f;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int ro = (e & 3) + 8 * (e >> 2);
            if (r0 + ro < R) wp[(int64_t)ro * C] = wv[e] + acc[e];
        }
    }
    }  // half
}

// evopress/src/quant_utils.py:57-106 Quantizer.find_params(x, weight=True), perchannel: one wave per row of the
// [R, G] panel at x (row stride ld): min / max over the group, the symmetric range, (-1, +1) for a constant row,
// scale = (max - min) / maxq, zero = round(-min / scale) or (maxq + 1) / 2.  G % 4 == 0, x 16-byte aligned.
__global__ __launch_bounds__(256) void uniform_params_kernel(const float* __restrict__ x, int64_t R, int64_t ld, int G,
                                                             float maxq, int sym, float* __restrict__ scale,
                                                             float* __restrict__ zero, int64_t ng) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const float* xr = x + row * ld;
    float mn = xr[0], mx = mn;
    for (int j = lane * 4; j < G; j += 256) {
        const float4 v = *reinterpret_cast<const float4*>(xr + j);
        mn = fminf(fminf(mn, v.x), fminf(v.y, fminf(v.z, v.w)));
        mx = fmaxf(fmaxf(mx, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o));
        mx = fmaxf(mx, __shfl_xor(mx, o));
    }
    if (lane) return;
    if (sym) {
        mx = fmaxf(fabsf(mn), mx);
        if (mn < 0.0f) mn = -mx;
    }
    if (mn == mx) {
        mn = -1.0f;
        mx = 1.0f;
    }
    const float sc = (mx - mn) / maxq;
    scale[row * ng] = sc;
    zero[row * ng] = sym ? (maxq + 1.0f) / 2.0f : rintf(-mn / sc);
}

// Block scratch maintenance for block_size > 128 (generic path, VALU):
//   Wblk[r, j] = Wblk[r, j] + (-Err[r, e0+i]) * U[a+i, cj0+j]   for i = 0..n_i-1, in order.
__global__ __launch_bounds__(256) void block_far_update_kernel(float* __restrict__ Wblk, int64_t ld_blk,
                                                               int64_t ncols, int64_t R,
                                                               const float* __restrict__ Err, int64_t ld_err,
                                                               int64_t e0, int n_i, const float* __restrict__ U,
                                                               int64_t C, int64_t a, int64_t cj0) {
    const int64_t total = R * ncols;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / ncols, j = t % ncols;
        float w = Wblk[r * ld_blk + j];
        const float* e = Err + r * ld_err + e0;
        for (int i = 0; i < n_i; ++i) w = w + (-e[i]) * U[(a + i) * C + cj0 + j];
        Wblk[r * ld_blk + j] = w;
    }
}

__global__ __launch_bounds__(256) void copy2d_kernel(float* __restrict__ dst, int64_t ld_dst,
                                                     const float* __restrict__ src, int64_t ld_src, int64_t R,
                                                     int64_t ncols) {
    const int64_t total = R * ncols;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / ncols, j = t % ncols;
        dst[r * ld_dst + j] = src[r * ld_src + j];
    }
}

// ------------------------------------------------------------- K6 trailing update
// W[:, c2:] -= Err[R,B] @ U[c1:c2, c2:]  (gptq.py:270) on the fp32 matrix cores: see
// gq_gemm32.hpp (MODE 0: k-ordered fma chain from 0, then one subtraction).
int launch_trailing_update(float* Cmat, int64_t ldc, const float* A, int64_t lda, const float* B, int64_t ldb,
                           int64_t M, int64_t N, int64_t K, hipStream_t st) {
    ProfScope ps(PT_TRAILING, st);
    return launch_gemm32<false, 0, false>(Cmat, ldc, A, lda, B, ldb, M, N, K, st);
}

// ------------------------------------------------------------- orchestration
// Look-ahead depth of the trailing update (blocks of 128 columns per super-block), see gptq_quantize.
constexpr int LA = 8;
constexpr int LA_B = 128;  // only this block size takes the look-ahead path (the chain length is a template constant)

// ---- the far update next to the column loop (single-Linear look-ahead pipeline) ----
// The far update of super-block s (everything beyond it) is one MFMA-bound GEMM; the column loop of super-block s+1
// is a string of small latency-bound launches that needs the far update on ITS OWN 1024 columns only.  So the far
// update is cut by column groups (g = 1024-column group index; F_s[g] = super-block s's update of group g):
//   caller's stream:  loop(s) | F_s[s+1] | loop(s+1) | F_{s+1}[s+2] | ...
//   helper stream:             F_s[s+2] , F_s[s+3..] | F_{s+1}[s+3] , F_{s+1}[s+4..] | ...
// F_s[s+1] waits for F_{s-1}[s+1] (an event after that small launch); the helper launches are PERSISTENT with a
// bounded number of workgroups, so the loop's kernels always find free CUs instead of queueing behind 56-us GEMM
// tiles.  Every element of W still sees ((w - E_0 U_0) - E_1 U_1) - ... in the same order: results are unchanged.
// The error buffer is doubled (the helper may still read super-block s's errors while the loop fills s+1's).
static std::atomic<int> g_far_enabled{1};
int far_helper_enable(int on) { return g_far_enabled.exchange(on ? 1 : 0); }
static bool far_async_shape(int64_t R, int64_t C, int64_t B, int la) {
    if (!g_far_enabled.load()) return false;
    // options are read per call: tests and A/B runs flip them inside one process
    const bool off = opt(OPT_far_sync) != 0;
    const int64_t max_rows = opt(OPT_far_async_max_rows), min_sb = opt(OPT_far_async_min_sb);
    return !off && B == LA_B && R % 128 == 0 && C % 128 == 0 && R <= max_rows && C >= min_sb * (int64_t)la * B;
}
// One helper stream per device, held by ONE call at a time: from the start of its enqueue until its last helper launch
// has finished on the device.  A call that finds it taken runs the one-stream schedule (same results) -- several
// chains funnelled through one in-order helper stream would wait for each other's GEMMs (measured: 103 -> 112 ms
// per block step with all seven Linears on it).
struct FarHelper {
    hipStream_t st = nullptr;
    hipEvent_t done = nullptr;  // recorded behind the holder's last helper launch
    bool enqueueing = false, recorded = false;
};
static std::mutex g_far_mu;
static FarHelper g_far[64];
static int far_helper_acquire(FarHelper** out) {
    *out = nullptr;
    int dev = 0;
    GQ_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_far_mu);
    FarHelper& h = g_far[dev & 63];
    if (!h.st) {
        // (Measured and dropped: a helper stream confined to 192 CUs by hipExtStreamCreateWithCUMask -- the first n mask
        // bits are n CUs, n/8 per XCD, profiles/micro/cu_mask_probe.hip -- with plain launches instead of persistent
        // ones: the 4096 x 14336 loop takes 20.4 ms against 12.5, a block step 109 ms against 97.)
        GQ_HIP(hipStreamCreateWithFlags(&h.st, hipStreamNonBlocking));
        GQ_HIP(hipEventCreateWithFlags(&h.done, hipEventDisableTiming));
    }
    if (h.enqueueing) return GQ_OK;
    if (h.recorded && hipEventQuery(h.done) != hipSuccess) {
        (void)hipGetLastError();  // hipErrorNotReady is not an error
        return GQ_OK;
    }
    h.enqueueing = true;
    *out = &h;
    return GQ_OK;
}
static void far_helper_release(FarHelper* h) {
    if (!h) return;
    std::lock_guard<std::mutex> lk(g_far_mu);
    h->recorded = hipEventRecord(h->done, h->st) == hipSuccess;
    h->enqueueing = false;
}
struct FarHold {  // releases on every return path of the column loop
    FarHelper* h = nullptr;
    ~FarHold() { far_helper_release(h); }
};
// re-recordable events of the calling host thread (a wait captures the record that precedes it)
static int far_event(int i, hipEvent_t* out) {
    thread_local hipEvent_t pool[96] = {};
    hipEvent_t& e = pool[i % 96];
    if (!e) GQ_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *out = e;
    return GQ_OK;
}

// does gq_gptq_quantize(R, C, block_size) put its far updates on the library's helper stream?
int gptq_uses_helper_stream(int64_t R, int64_t C, int block_size) {
    const int64_t B = block_size <= 0 || block_size > C ? C : block_size;
    int la = LA;
    if (const int64_t e = opt(OPT_la)) la = (e >= 2 && e <= LA && e % 2 == 0) ? (int)e : LA;
    return !opt(OPT_no_lookahead) && far_async_shape(R, C, B, la);
}

size_t gptq_workspace_bytes(int64_t R, int64_t C, int block_size) {
    int64_t B = block_size <= 0 || block_size > C ? C : block_size;
    size_t err = (size_t)R * (size_t)B * sizeof(float);
    if (B == LA_B) err *= LA;  // super-block error buffer [R, LA * B]
    if (B == LA_B) err *= 2;   // ... doubled for the far update next to the loop (far_async_shape)
    size_t blk = B > SEG ? (size_t)R * (size_t)B * sizeof(float) : 0;
    return err + blk + 256 + 256;  // + the panel word of the scale searches (quant_utils.py:250-252)
}

// perm != nullptr: act_order (gptq.py:208-216, 233-235, 272-276).  W and U are already in permuted
// order, d/s/dmin/m are INPUTS (the static scales of the original column groups, gptq.py:184-196) and
// qweight comes back in permuted positions; the caller un-permutes it.
//
// uni != nullptr: the uniform grids of EvoPress' FastOBQ (evopress/src/fast_obq.py:146-200) instead of a K-quant --
// the same column loop and trailing updates; the grid of a group is found from W as it is when the block holding
// the group's first column starts (fast_obq.py:168-171 reads w, which the in-block feedback never touches).
struct UniformSpec {
    int bits, group, sym;  // group == 0: one grid per row, from the original W (fast_obq.py:153-154)
    float *scale, *zero;   // [R, C / group]
};

static int column_loop(float* W, const float* U, int64_t R, int64_t C, int q_type, int block_size, int static_groups,
                       const gq_search_t* p, uint8_t* qweight, uint16_t* d, uint8_t* s, uint16_t* dmin, uint8_t* m,
                       void* ws, size_t ws_bytes, hipStream_t st, const int32_t* perm, const UniformSpec* uni,
                       const int64_t* row_ends = nullptr, int nstack = 1, int32_t* researches_out = nullptr) {
    // row_ends / nstack: W is several matrices that share U, one under the other (rows never mix: gptq.py:222-270);
    // only the scale searches need to know where one ends and the next begins (their panel-wide `continue`)
    TypeInfo ti;
    if (uni) {
        ti = TypeInfo{};
        ti.group = uni->group > 0 ? uni->group : (int)C;
        ti.is_signed = 0;
        ti.qmin = 0;
        ti.qmax = (1 << uni->bits) - 1;
        ti.k_search = false;
        static_groups = 2;  // no K-quant scale search
        d = dmin = reinterpret_cast<uint16_t*>(qweight);  // unused by the UNI kernel; keeps the null check below simple
        s = m = qweight;
    } else if (!type_info(q_type, ti)) GQ_FAIL(GQ_E_BAD_TYPE, "gq_gptq_quantize: unknown q_type %d", q_type);
    if (R <= 0 || C <= 0 || C % 256) GQ_FAIL(GQ_E_BAD_SHAPE, "gq_gptq_quantize: R=%ld C=%ld (C %% 256 != 0)", (long)R, (long)C);
    if (!W || !U || !qweight || !d || !s || !dmin || !m) GQ_FAIL(GQ_E_NULL, "gq_gptq_quantize: null pointer");
    const int64_t B = block_size <= 0 || block_size > C ? C : block_size;  // gptq.py:54
    int rc;
    if (B % SB) GQ_FAIL(GQ_E_UNSUPPORTED, "gq_gptq_quantize: block_size %ld is not a multiple of 16", (long)B);
    if (q_type == GQ_Q3_K && perm) GQ_FAIL(GQ_E_UNSUPPORTED, "gq_gptq_quantize_perm: Q3_K forces act_order off (gptq.py:204-206)");
    if (q_type == GQ_Q3_K) static_groups = 0;  // gptq.py:204-206
    if (perm) static_groups = 2;  // scales are inputs: neither the up-front nor the lazy search runs
    const size_t need = gptq_workspace_bytes(R, C, (int)B);
    if (!ws || ws_bytes < need) GQ_FAIL(GQ_E_WORKSPACE, "gq_gptq_quantize: workspace %zu < %zu bytes", ws_bytes, need);
    // Look-ahead trailing update (B == 128): the blocks of a super-block of LA blocks write their errors
    // side by side into one [R, LA*128] buffer.  After each block only the REST OF THE SUPER-BLOCK is
    // updated (a small GEMM); at the end of the super-block everything beyond it is updated by ONE chained
    // GEMM (CHAIN = 128): per element ((w - E_0 U_0) - E_1 U_1) - ..., the same operations in the same
    // order as gptq.py:270 applied block after block, with one read and one write of W instead of LA.
    // (Deferring the far part to a helper stream, to overlap it with the next super-block's column loop,
    // measured no gain alone and -8 % inside a block's multi-stream schedule: its long K = 1024 tiles keep
    // the column-loop workgroups, which need a whole CU's LDS, waiting.)
    int la = LA;
    // tuning knob; even only: a 256-column scale-search group must not straddle two super-blocks
    if (const int64_t e = opt(OPT_la)) la = (e >= 2 && e <= LA && e % 2 == 0) ? (int)e : LA;
    // a uniform group must lie inside one super-block as well: its grid is found from columns that every earlier
    // block has already updated
    const bool lookahead = (B == LA_B) && !opt(OPT_no_lookahead) &&
                           (!uni || uni->group <= 0 || (la * B) % uni->group == 0);
    // (Measured and removed, r03: a LEFT-looking form of the near updates -- before block b its columns receive the errors of
    // the super-block's earlier blocks in one chained 64-tile launch with K = 128 b: a quarter of the traffic on W, bit-identical,
    // and no faster (2.33 vs 2.30 ms of near updates in the 4096 x 14336 loop))
    // r03, default: near updates at the granularity of the 256-column scale-search groups.  An even block hands its
    // errors to its partner (the odd block of the group) in the column-loop kernel's epilogue; after the odd block ONE
    // chained launch (K = 256, chain 128) brings both blocks' errors to the rest of the super-block.  Per element the
    // same subtractions in the same order as after-every-block updates (bit-identical: the parity tests run both);
    // 3 launches per super-block instead of 7.  Option near_classic: one launch after every block (r02).
    const bool pair_look = lookahead && !opt(OPT_near_classic) && la % 2 == 0 &&
                           (!uni || uni->group <= 0 || 256 % uni->group == 0);
    const int64_t ldE = lookahead ? (int64_t)LA * B : B;
    float* Err0 = reinterpret_cast<float*>(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    float* Wblk = Err0 + (size_t)R * B * (B == LA_B ? 2 * LA : 1);
    bool far_async = lookahead && far_async_shape(R, C, B, la);
    const int far_wgs = (int)opt(OPT_far_wgs);
    FarHold hold;
    hipStream_t helper = nullptr;
    hipEvent_t ev_small_prev = nullptr, ev_bulk[2] = {nullptr, nullptr}, ev_last = nullptr;
    if (far_async) {
        if ((rc = far_helper_acquire(&hold.h))) return rc;
        if (hold.h) helper = hold.h->st;
        else far_async = false;  // taken by another call: the one-stream schedule
    }
    int ev_i = 0;
    // one device word shared by all scale-search launches of this call (each leaves it at zero)
    unsigned* panel = reinterpret_cast<unsigned*>(
        (reinterpret_cast<uintptr_t>(Wblk + ((B > SEG) ? (size_t)R * B : 0)) + 255) & ~(uintptr_t)255);
    if ((ti.k_search && static_groups != 2) || researches_out) GQ_HIP(hipMemsetAsync(panel, 0, 256, st));
    const int64_t ng = C / ti.group, nsg = C / 256;
    const int gps = uni ? 1 : 256 / ti.group;
    auto uniform_params = [&](int64_t col, int G, int64_t g) {
        hipLaunchKernelGGL(uniform_params_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, W + col, R, C, G,
                           (float)ti.qmax, uni->sym, uni->scale + g, uni->zero + g, ng);
    };
    if (uni && uni->group <= 0) {
        uniform_params(0, (int)C, 0);
        GQ_LAUNCH_CHECK();
    }

    if (static_groups == 1) {  // gptq.py:184-196: all scales from the ORIGINAL W
        for (int64_t c = 0; c < C; c += 256)
            if ((rc = launch_scale_search(W + c, R, C, q_type, p, d + c / 256, nsg, s + (c / 256) * gps, ng,
                                          dmin + c / 256, nsg, m + (c / 256) * gps, ng, st, panel, row_ends, nstack)))
                return rc;
    }
    const dim3 seg_grid((unsigned)((R + 63) / 64)), seg_block(SEG_WAVES * 64);
    static std::atomic<bool> seg_attr{false};  // guards an idempotent call: a race sets the same value twice
    if (!seg_attr) {
        GQ_HIP(hipFuncSetAttribute((const void*)gptq_segment_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   SEG_LDS_BYTES));
        GQ_HIP(hipFuncSetAttribute((const void*)gptq_segment_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   SEG_LDS_BYTES));
        GQ_HIP(hipFuncSetAttribute((const void*)gptq_segment_kernel<false, true>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, SEG_LDS_BYTES));
        seg_attr = true;
    }
    const bool seg_pair = opt(OPT_seg_pair) != 0;
    bool walked_by_partner = false;  // this block's columns were walked by the previous (even) block's launch
    for (int64_t c1 = 0; c1 < C; c1 += B) {  // gptq.py:222
        const int64_t c2 = c1 + B < C ? c1 + B : C;
        // one segment iff the block fits in LDS and stays inside one 256-column super-group
        const bool single = (c2 - c1) <= SEG && (c1 / 256 == (c2 - 1) / 256);
        const int64_t ncols = c2 - c1;
        const int64_t bi = c1 / B, sb = bi / la, pos = lookahead ? bi % la : 0;  // block, super-block, slot
        float* Err = far_async ? Err0 + (sb & 1) * (size_t)R * ldE : Err0;
        if (far_async && pos == 0 && ev_bulk[sb & 1]) {  // this half of the error buffer: the helper is done with it
            GQ_HIP(hipStreamWaitEvent(st, ev_bulk[sb & 1], 0));
            ev_bulk[sb & 1] = nullptr;
        }
        if (uni && uni->group > 0) {  // fast_obq.py:168-171 for every group that starts inside this block
            ProfScope ps(PT_SCALE_SEARCH, st);
            for (int64_t g = (c1 + uni->group - 1) / uni->group; g * uni->group < c2; ++g) uniform_params(g * uni->group, uni->group, g);
            GQ_LAUNCH_CHECK();
        }
        if (!single) {  // w_blk lives in scratch
            ProfScope ps(PT_BLOCK_FAR, st);
            hipLaunchKernelGGL(copy2d_kernel, dim3(2048), dim3(256), 0, st, Wblk, B, W + c1, C, R, ncols);
            GQ_LAUNCH_CHECK();
        }
        int64_t a = walked_by_partner ? c2 : c1;
        walked_by_partner = false;
        while (a < c2) {
            // a segment never crosses a 256-column super-group boundary: the lazy
            // scale search (gptq.py:240-245) must see W as it is at that column.
            int64_t e = a + SEG < c2 ? a + SEG : c2;
            const int64_t next_sg = (a / 256 + 1) * 256;
            if (e > next_sg) e = next_sg;
            const int len = (int)(e - a);
            if (!static_groups && (a % 256) == 0) {
                // reads w (global), NOT w_blk: with block_size > 256 these columns
                // are stale by design
                const int64_t sg = a / 256;
                if ((rc = launch_scale_search(W + a, R, C, q_type, p, d + sg, nsg, s + sg * gps, ng, dmin + sg, nsg,
                                              m + sg * gps, ng, st, panel, row_ends, nstack)))
                    return rc;
            }
            // an even block of a 256-group: its partner's columns are updated in this kernel's epilogue
            const float* unext = (pair_look && single && len == SEG && !(pos & 1) && c2 + B <= C) ? U + a * C + a + SEG : nullptr;
            const float* srcp = single ? (W + a) : (Wblk + (a - c1));
            const int64_t ld_src = single ? C : B;
            // the partner block in the same launch (it is a whole single segment too: B == SEG, c2 + B <= C)
            const int npair = (unext && seg_pair && !uni && B == SEG) ? 2 : 1;
            walked_by_partner = npair == 2;
            {
                ProfScope ps(PT_GPTQ_SEGMENT, st);
                if (uni)
                    hipLaunchKernelGGL((gptq_segment_kernel<false, true>), seg_grid, seg_block, SEG_LDS_BYTES, st, W, C, srcp,
                                       ld_src, U, a, len, R, d, s, dmin, m, ti.group, 0, 0.0f, (float)ti.qmax, qweight, Err,
                                       ldE, pos * B + (a - c1), perm, uni->scale, uni->zero, unext);
                else if (perm)
                    hipLaunchKernelGGL(gptq_segment_kernel<true>, seg_grid, seg_block, SEG_LDS_BYTES, st, W, C, srcp, ld_src, U, a,
                                       len, R, d, s, dmin, m, ti.group, ti.is_signed, (float)ti.qmin, (float)ti.qmax, qweight, Err,
                                       ldE, pos * B + (a - c1), perm, nullptr, nullptr, unext, npair);
                else
                    hipLaunchKernelGGL(gptq_segment_kernel<false>, seg_grid, seg_block, SEG_LDS_BYTES, st, W, C, srcp, ld_src, U, a,
                                       len, R, d, s, dmin, m, ti.group, ti.is_signed, (float)ti.qmin, (float)ti.qmax, qweight, Err,
                                       ldE, pos * B + (a - c1), perm, nullptr, nullptr, unext, npair);
                GQ_LAUNCH_CHECK();
            }
            if (e < c2) {  // push this segment's rank-1 updates into the rest of the block
                ProfScope ps(PT_BLOCK_FAR, st);
                hipLaunchKernelGGL(block_far_update_kernel, dim3(2048), dim3(256), 0, st, Wblk + (e - c1), B,
                                   c2 - e, R, Err, ldE, a - c1, len, U, C, a, e);
                GQ_LAUNCH_CHECK();
            }
            a = e;
        }
        // gptq.py:270
        if (c2 >= C) break;
        if (!lookahead) {
            if ((rc = launch_trailing_update(W + c2, C, Err, B, U + c1 * C + c2, C, R, C - c2, ncols, st))) return rc;
            continue;
        }
        const int64_t S0 = sb * la * B, S1 = (S0 + la * B < C) ? S0 + la * B : C;  // this super-block
        if (c2 < S1) {  // rest of the super-block, this block's errors only
            // (measured and dropped: applying the blocks in pairs -- an even block updates its partner only, the odd
            // block the rest with both in one chained K = 256 launch -- gives 4 tiny + 3 deeper launches instead of 7
            // thin ones, bit-identical, but no faster: 1.79 vs 1.71 ms for 4096 x 14336; every launch is latency.
            // Also dropped: a dedicated K = 128 kernel, one workgroup per CU with the whole K of both operands in LDS
            // (23.5 vs 17.3 us per launch).  A rank-128 update moves 16 B of operands L2 -> LDS per output element for
            // 256 flops at 64 x 64 tiles: it is L2-bandwidth-bound near 50 TFLOP/s whatever the schedule.)
            if (pair_look) {
                // GQ_NEAR_QUAD=1 (measured, off): a third level -- pair -> the quad's other pair (K = 256, N = 256), quad -> the
                // rest of the super-block (K = 512, N = 512): same flops, deeper K on fewer tiles: 1.24 vs 1.05 ms of near
                // launches for 4096 x 14336
                const bool quad = opt(OPT_near_quad) != 0;
                if ((pos & 1) && !quad) {  // end of a 256-group: both blocks' errors, in order, to the rest of the super-block
                    ProfScope ps(PT_TRAILING, st);
                    // up to GQ_NEAR64_MAXN columns: 64x64 tiles with the K = 256 panels whole in LDS (gemm32_near256_kernel)
                    const int64_t near64_maxn = opt(OPT_near64_maxn);
                    if (S1 - c2 <= near64_maxn && R % 64 == 0) {
                        if ((rc = launch_gemm32_near256(W + c2, C, Err + (pos - 1) * B, ldE, U + (c1 - B) * C + c2, C, R, S1 - c2, st)))
                            return rc;
                    } else if ((rc = launch_gemm32<false, 0, false, 0, LA_B>(W + c2, C, Err + (pos - 1) * B, ldE,
                                                                             U + (c1 - B) * C + c2, C, R, S1 - c2, 2 * B, st)))
                        return rc;
                } else if ((pos & 3) == 1) {  // first pair of a 512-column quad: its two blocks' errors to the quad's other pair
                    const int64_t n = (c2 + 2 * B < S1) ? 2 * B : S1 - c2;
                    ProfScope ps(PT_TRAILING, st);
                    if ((rc = launch_gemm32<false, 0, false, 0, LA_B>(W + c2, C, Err + (pos - 1) * B, ldE, U + (c1 - B) * C + c2, C, R, n,
                                                                      2 * B, st)))
                        return rc;
                } else if ((pos & 3) == 3) {  // end of a quad: its four blocks' errors, in order, to the rest of the super-block
                    ProfScope ps(PT_TRAILING, st);
                    if ((rc = launch_gemm32<false, 0, false, 0, LA_B>(W + c2, C, Err + (pos - 3) * B, ldE, U + (c1 - 3 * B) * C + c2, C, R,
                                                                      S1 - c2, 4 * B, st)))
                        return rc;
                }
                continue;
            }
            if ((rc = launch_trailing_update(W + c2, C, Err + pos * B, ldE, U + c1 * C + c2, C, R, S1 - c2, ncols, st)))
                return rc;
            continue;
        }
        // end of the super-block: all its blocks at once, every later column
        if (!far_async) {
            ProfScope ps(PT_TRAILING_FAR, st);
            if ((rc = launch_gemm32<false, 0, false, 0, LA_B>(W + S1, C, Err, ldE, U + S0 * C + S1, C, R, C - S1, S1 - S0, st)))
                return rc;
            continue;
        }
        {
            const int64_t G = (int64_t)la * B, g1 = S1 + G < C ? S1 + G : C, g2 = g1 + G < C ? g1 + G : C;
            hipEvent_t ready = nullptr, ev = nullptr;
            if (g1 < C) {  // the helper may start on this super-block's errors
                if ((rc = far_event(ev_i++, &ready))) return rc;
                GQ_HIP(hipEventRecord(ready, st));
            }
            if (ev_small_prev) GQ_HIP(hipStreamWaitEvent(st, ev_small_prev, 0));  // F_{s-1}[s+1] is in
            ev_small_prev = nullptr;
            {
                ProfScope ps(PT_TRAILING_FAR, st);
                if ((rc = launch_gemm32_chain_full<LA_B>(W + S1, C, Err, ldE, U + S0 * C + S1, C, R, g1 - S1, S1 - S0, st)))
                    return rc;
            }
            if (g1 < C) {
                GQ_HIP(hipStreamWaitEvent(helper, ready, 0));
                ProfScope ps(PT_TRAILING_FAR, helper);
                if ((rc = launch_gemm32_chain_full<LA_B>(W + g1, C, Err, ldE, U + S0 * C + g1, C, R, g2 - g1, S1 - S0, helper, far_wgs)))
                    return rc;
                if ((rc = far_event(ev_i++, &ev))) return rc;
                GQ_HIP(hipEventRecord(ev, helper));
                ev_small_prev = ev;
                if (g2 < C && (rc = launch_gemm32_chain_full<LA_B>(W + g2, C, Err, ldE, U + S0 * C + g2, C, R, C - g2, S1 - S0, helper, far_wgs)))
                    return rc;
                if ((rc = far_event(ev_i++, &ev))) return rc;
                GQ_HIP(hipEventRecord(ev, helper));
                ev_bulk[sb & 1] = ev;
                ev_last = ev;
            }
        }
    }
    if (ev_last) GQ_HIP(hipStreamWaitEvent(st, ev_last, 0));  // the caller's stream sees the helper's last write
    if (researches_out)  // every scale search of this call ran on `st`: the count is final here
        GQ_HIP(hipMemcpyAsync(researches_out, panel + GQ_PANEL_RESEARCH, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    return GQ_OK;
}

int gptq_quantize(float* W, const float* U, int64_t R, int64_t C, int q_type, int block_size, int static_groups,
                  const gq_search_t* p, uint8_t* qweight, uint16_t* d, uint8_t* s, uint16_t* dmin, uint8_t* m,
                  void* ws, size_t ws_bytes, hipStream_t st, const int32_t* perm, const int64_t* row_ends, int nstack,
                  int32_t* researches_out) {
    if (nstack > 1 && perm) GQ_FAIL(GQ_E_UNSUPPORTED, "gq_gptq_quantize_stacked: act_order matrices are not stacked");
    return column_loop(W, U, R, C, q_type, block_size, static_groups, p, qweight, d, s, dmin, m, ws, ws_bytes, st, perm,
                       nullptr, row_ends, nstack, researches_out);
}

// EvoPress FastOBQ.step for one bit width (evopress/src/fast_obq.py:146-200) given U.
int obq_quantize(float* W, const float* U, int64_t R, int64_t C, int bits, int group_size, int sym, int block_size,
                 uint8_t* qweight, float* scale, float* zero, void* ws, size_t ws_bytes, hipStream_t st) {
    if (bits < 1 || bits > 8) GQ_FAIL(GQ_E_UNSUPPORTED, "gq_obq_quantize: bits=%d (1..8: qweight is uint8)", bits);
    if (!scale || !zero) GQ_FAIL(GQ_E_NULL, "gq_obq_quantize: null pointer");
    if (group_size < 0 || (group_size > 0 && (group_size % SB || C % group_size)))
        GQ_FAIL(GQ_E_UNSUPPORTED, "gq_obq_quantize: group_size %d must be 0 or a multiple of 16 that divides C=%ld", group_size, (long)C);
    UniformSpec u{bits, group_size, sym ? 1 : 0, scale, zero};
    return column_loop(W, U, R, C, -1, block_size, 2, nullptr, qweight, nullptr, nullptr, nullptr, nullptr, ws, ws_bytes,
                       st, nullptr, &u);
}

}  // namespace gq
