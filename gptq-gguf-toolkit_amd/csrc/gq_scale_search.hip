// gq_scale_search.hip -- K4: per-super-group scale/min search (get_scale_and_zero).
//
// Reference: quant_utils.py:90-145 (outer), :199-274 make_k_quants (Q2/Q4/Q5),
// :147-197 make_quants (Q3/Q6).  Bit-exact with the reference's CPU path given the
// same fp32 panel; every quirk is kept on purpose (cited inline).
//
// Mapping (CDNA4, wave64).  ATen's CPU inner-dim sum adds a 16/32-element group
// as EIGHT lane accumulators (element e -> accumulator e%8, in order of e/8) that
// are then summed 0 -> 7.  The kernel uses exactly that shape: 8 lanes per group,
// lane l holds x[l], x[l+8], (x[l+16], x[l+24]); the lane accumulators are the
// per-lane partial sums, and the ordered 0 -> 7 sum is a 7-step DPP row_shr:1
// running sum whose result lands on lane 7 and is re-broadcast with one
// ds_swizzle.  One 256-thread workgroup = 4 row-panels (G=32) or 2 (G=16); the
// per-row amax/6-bit re-quantisation of the group scales goes through 128..256 B
// of LDS.  HBM traffic is one coalesced read of the fp32 panel (1 KiB per row)
// and ~20 B of outputs per row; the search itself is ~300 VALU-flop/param.
#include "gq_common.hpp"

namespace gq {

template <int CTRL>
__device__ __forceinline__ float dpp_f(float old, float src) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL,
                                           0xf, 0xf, false));
}

// sum of the 8 lane partials of an 8-lane group, added in lane order 0 -> 7
// (== gqo_aten_sum's final loop), result valid on EVERY lane of the group.
__device__ __forceinline__ float group_sum8(float partial) {
    float acc = partial;
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        // lane i reads lane i-1's running sum (row_shr:1).  After step k lane k
        // holds partial_0 + ... + partial_k added left to right; other lanes hold
        // values nobody reads.
        float prev = dpp_f<0x111>(acc, acc);
        acc = prev + partial;
    }
    // broadcast lane 7 of each 8-lane group: swizzle bit-mode, and=0x18 or=0x07
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, acc), 0x18 | (0x07 << 5)));
}

__device__ __forceinline__ float group_min8(float v) {
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        float t = __shfl_xor(v, o);
        v = t < v ? t : v;
    }
    return v;
}
__device__ __forceinline__ float group_max8(float v) {
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        float t = __shfl_xor(v, o);
        v = t > v ? t : v;
    }
    return v;
}

struct SearchParams {
    float num[24];  // fp32(rmin + rdelta*i + maxq), evaluated in double on the host
    int nstep;
};

// quant_utils.py:199-274 for one group spread over 8 lanes; NS = G/8 values per lane.
template <int NS, int BITS>
__device__ __forceinline__ void k_search(const float (&x)[NS], const SearchParams& sp, float& scale_out,
                                         float& zero_out) {
    constexpr float maxq = (float)((1 << BITS) - 1);
    constexpr float G = (float)(NS * 8);
    const float eps = 1e-9f;
    float w[NS];
    // :203-205
    float p = x[0] * x[0];
#pragma unroll
    for (int k = 1; k < NS; ++k) p = p + x[k] * x[k];
    float sum_x2 = group_sum8(p);
    float av_x = sqrtf(sum_x2 / G);  // IEEE sqrt (MKL vsSqrt note)
#pragma unroll
    for (int k = 0; k < NS; ++k) w[k] = av_x + fabsf(x[k]);
    // :208-211
    float mn = x[0], mx = x[0];
#pragma unroll
    for (int k = 1; k < NS; ++k) {
        mn = x[k] < mn ? x[k] : mn;
        mx = x[k] > mx ? x[k] : mx;
    }
    mn = group_min8(mn);
    mx = group_max8(mx);
    mn = mn < 0.0f ? mn : 0.0f;
    float x_min = mn;
    const float x_max = mx;
    const bool is_const = (x_max == x_min);
    // :214-215
    float pw = w[0], px = w[0] * x[0];
#pragma unroll
    for (int k = 1; k < NS; ++k) {
        pw = pw + w[k];
        px = px + w[k] * x[k];
    }
    const float sum_w = group_sum8(pw);
    const float sum_x = group_sum8(px);
    // :218-232
    float sc = (x_max - x_min) / maxq;
    if (is_const) sc = 0.0f;
    const float isc = 1.0f / (sc < eps ? eps : sc);
    float pe;
    {
        float e[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            float q = clampf(rintf((x[k] - x_min) * isc), 0.0f, maxq);
            if (is_const) q = 0.0f;
            float diff = (sc * q + x_min) - x[k];
            e[k] = w[k] * (diff * diff);
        }
        pe = e[0];
#pragma unroll
        for (int k = 1; k < NS; ++k) pe = pe + e[k];
    }
    float best_err = group_sum8(pe);
    float best_scale = sc;

    if (sp.nstep >= 1) {  // :235-237
        for (int i = 0; i <= sp.nstep; ++i) {  // :240
            // :241 scalar/tensor == reciprocal()*scalar; x_min is the aliased best_min (:228,:270)
            float den = x_max - x_min;
            den = den < eps ? eps : den;
            const float cand_iscale = (1.0f / den) * sp.num[i];
            float L[NS];
            float pl, pl2, pxl;
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                float q = clampf(rintf((x[k] - x_min) * cand_iscale), 0.0f, maxq);  // :242
                if (is_const) q = 0.0f;                                            // :243
                int qi = (int)q;
                float q2 = (float)((qi * qi) & 255);  // :246 new_q**2 stays uint8 (wraps for Q5_K)
                L[k] = q;
                float tl = w[k] * q, tl2 = w[k] * q2, txl = (w[k] * x[k]) * q;
                if (k == 0) {
                    pl = tl; pl2 = tl2; pxl = txl;
                } else {
                    pl = pl + tl; pl2 = pl2 + tl2; pxl = pxl + txl;
                }
            }
            const float sum_l = group_sum8(pl);
            const float sum_l2 = group_sum8(pl2);
            const float sum_xl = group_sum8(pxl);
            const float D = sum_w * sum_l2 - sum_l * sum_l;               // :249
            float this_scale = (sum_w * sum_xl - sum_x * sum_l) / D;     // :254
            float this_min = (sum_l2 * sum_x - sum_l * sum_xl) / D;      // :255
            if (this_min > 0.0f) {                                       // :257-260
                this_scale = sum_xl / (sum_l2 < eps ? eps : sum_l2);
                this_min = 0.0f;
            }
            float pc;
#pragma unroll
            for (int k = 0; k < NS; ++k) {  // :262-264
                float diff = (this_scale * L[k] + this_min) - x[k];
                float e = w[k] * (diff * diff);
                pc = (k == 0) ? e : pc + e;
            }
            const float cand_err = group_sum8(pc);
            // :250-252 the panel-wide `if not valid.any(): continue` is NOT taken
            // here: it only differs when EVERY group of the [rows,256] panel has
            // D <= 1e-9 in this iteration (|x| <~ 1e-7 everywhere); see DESIGN.md.
            if (cand_err < best_err) {  // :266-271 (NaN compares false)
                best_err = cand_err;
                best_scale = this_scale;
                x_min = this_min;
            }
        }
    }
    scale_out = best_scale;
    zero_out = -x_min;  // :273
}

// quant_utils.py:147-197 (absmax branch)
template <int NS, int BITS>
__device__ __forceinline__ void absmax_search(const float (&x)[NS], float& scale_out, float& zero_out) {
    constexpr float maxq = (float)((1 << BITS) - 1);
    float mn = x[0], mx = x[0];
#pragma unroll
    for (int k = 1; k < NS; ++k) {
        mn = x[k] < mn ? x[k] : mn;
        mx = x[k] > mx ? x[k] : mx;
    }
    mn = group_min8(mn);
    mx = group_max8(mx);
    float a = fabsf(mn);
    mx = a > mx ? a : mx;     // :153
    if (mn < 0.0f) mn = -mx;  // :154-156
    if (mn == mx) {           // :157-159
        mn = -1.0f;
        mx = 1.0f;
    }
    scale_out = (mx - mn) / maxq;  // :161
    zero_out = 0.0f;               // :195
}

// One workgroup = 256 threads = 256/(NG*8) row-panels.  NG = 256/G groups per row.
template <int GSZ, int BITS, bool KSEARCH, bool SIGNED, int SMQ>
__global__ __launch_bounds__(256) void scale_search_kernel(
    const float* __restrict__ x, int64_t rows, int64_t ld, SearchParams sp,
    uint16_t* __restrict__ d, int64_t d_stride, uint8_t* __restrict__ s, int64_t s_ld,
    uint16_t* __restrict__ dmin, int64_t dmin_stride, uint8_t* __restrict__ m, int64_t m_ld) {
    constexpr int NS = GSZ / 8;
    constexpr int NG = 256 / GSZ;
    constexpr int LPR = NG * 8;         // lanes per row
    constexpr int RPW = 256 / LPR;      // rows per workgroup
    __shared__ float sh_scale[RPW][NG];
    __shared__ float sh_zero[RPW][NG];

    const int tid = threadIdx.x;
    const int row_l = tid / LPR;
    const int g = (tid % LPR) / 8;
    const int l8 = tid & 7;
    const int64_t row = (int64_t)blockIdx.x * RPW + row_l;
    const bool live = row < rows;
    const float* xr = x + (live ? row : 0) * ld + g * GSZ + l8;
    float xv[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) xv[k] = xr[k * 8];

    float gscale, gzero;
    if constexpr (KSEARCH) k_search<NS, BITS>(xv, sp, gscale, gzero);
    else absmax_search<NS, BITS>(xv, gscale, gzero);
    if (l8 == 0) {
        sh_scale[row_l][g] = gscale;
        sh_zero[row_l][g] = gzero;
    }
    __syncthreads();
    if (l8 == 0 && live) {
        // quant_utils.py:121-143
        float max_scale = sh_scale[row_l][0], max_zero = sh_zero[row_l][0];
#pragma unroll
        for (int j = 1; j < NG; ++j) {
            float a = sh_scale[row_l][j], b = sh_zero[row_l][j];
            max_scale = a > max_scale ? a : max_scale;
            max_zero = b > max_zero ? b : max_zero;
        }
        constexpr float smq = (float)SMQ;
        float inv_scale = max_scale > 0.0f ? (1.0f / max_scale) * smq : 0.0f;  // :128
        float inv_zero = max_zero > 0.0f ? (1.0f / max_zero) * smq : 0.0f;     // :129
        float a = clampf(rintf(inv_scale * gscale), 0.0f, smq);                // :132-143
        float b = clampf(rintf(inv_zero * gzero), 0.0f, smq);
        s[row * s_ld + g] = SIGNED ? (uint8_t)(int8_t)a : (uint8_t)a;
        m[row * m_ld + g] = SIGNED ? (uint8_t)(int8_t)b : (uint8_t)b;
        if (g == 0) {
            d[row * d_stride] = f2h(max_scale / smq);    // :124
            dmin[row * dmin_stride] = f2h(max_zero / smq);  // :125
        }
    }
}

int launch_scale_search(const float* x, int64_t rows, int64_t ld, int q_type, const gq_search_t* p,
                        uint16_t* d, int64_t d_stride, uint8_t* s, int64_t s_ld, uint16_t* dmin,
                        int64_t dmin_stride, uint8_t* m, int64_t m_ld, hipStream_t st) {
    TypeInfo ti;
    if (!type_info(q_type, ti)) GQ_FAIL(GQ_E_BAD_TYPE, "gq_scale_search: unknown q_type %d", q_type);
    if (rows <= 0 || ld < 256) GQ_FAIL(GQ_E_BAD_SHAPE, "gq_scale_search: rows=%ld ld=%ld", (long)rows, (long)ld);
    SearchParams sp;
    sp.nstep = p ? p->nstep : 20;
    if (sp.nstep > 23) GQ_FAIL(GQ_E_UNSUPPORTED, "gq_scale_search: nstep=%d > 23", sp.nstep);
    const double rmin = p ? p->rmin : -1.0, rdelta = p ? p->rdelta : 0.1;
    const double maxq = (double)((1 << ti.bits) - 1);
    for (int i = 0; i < 24; ++i) sp.num[i] = (float)(rmin + rdelta * (double)i + maxq);
    const int rpw = ti.group == 32 ? 4 : 2;
    dim3 grid((unsigned)((rows + rpw - 1) / rpw)), block(256);
    ProfScope ps(PT_SCALE_SEARCH, st);
#define GQ_SS(G, B, K, S, Q)                                                                                \
    hipLaunchKernelGGL((scale_search_kernel<G, B, K, S, Q>), grid, block, 0, st, x, rows, ld, sp, d, d_stride, \
                       s, s_ld, dmin, dmin_stride, m, m_ld)
    switch (q_type) {
    case GQ_Q2_K: GQ_SS(16, 2, true, false, 15); break;
    case GQ_Q3_K: GQ_SS(16, 3, false, true, 31); break;
    case GQ_Q4_K: GQ_SS(32, 4, true, false, 63); break;
    case GQ_Q5_K: GQ_SS(32, 5, true, false, 63); break;
    case GQ_Q6_K: GQ_SS(16, 6, false, true, 63); break;
    }
#undef GQ_SS
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

// RTN scale search in the model dtype (quantizer.py:109,195 hand module.weight to
// get_scale_and_zero un-cast, so every op of make_*quants rounds to fp16/bf16).
int launch_rtn_scale_search(const void* W, int w_dtype, int64_t R, int64_t C, int q_type, const gq_search_t* p,
                            uint16_t* d, uint8_t* s, uint16_t* dmin, uint8_t* m, hipStream_t st) {
    (void)W; (void)R; (void)C; (void)q_type; (void)p; (void)d; (void)s; (void)dmin; (void)m; (void)st;
    GQ_FAIL(GQ_E_UNSUPPORTED,
            "gq_rtn_quantize: w_dtype %d (reduced-precision make_*quants emulation) is not implemented; "
            "pass the weight as fp32 (reference behaviour with --dtype float32)", w_dtype);
}

}  // namespace gq
