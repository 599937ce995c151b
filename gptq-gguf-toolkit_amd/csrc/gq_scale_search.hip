// gq_scale_search.hip -- K4: per-super-group scale/min search (get_scale_and_zero).
//
// Reference: quant_utils.py:90-145 (outer), :199-274 make_k_quants (Q2/Q4/Q5),
// :147-197 make_quants (Q3/Q6).  Bit-exact with the reference's CPU path given the
// same panel; every quirk is kept on purpose (cited inline).
//
// Mapping (CDNA4, wave64).  ATen's CPU inner-dim sum adds a 16/32-element group
// as EIGHT lane accumulators (element e -> accumulator e%8, in order of e/8) that
// are then summed 0 -> 7.  The kernel uses exactly that shape: 8 lanes per group,
// lane l holds x[l], x[l+8], (x[l+16], x[l+24]); the lane accumulators are the
// per-lane partial sums, and the ordered 0 -> 7 sum is a 7-step DPP row_shr:1
// running sum whose result lands on lane 7 and is re-broadcast with one
// ds_swizzle.  One 256-thread workgroup = 4 row-panels (G=32) or 2 (G=16); the
// per-row amax/6-bit re-quantisation of the group scales goes through 128..256 B
// of LDS.  HBM traffic is one coalesced read of the panel (1 KiB per row in fp32)
// and ~20 B of outputs per row; the search itself is ~300 VALU-flop/param.
//
// RM (rounding mode) 0: fp32 panel (GPTQ.step).  RM 1 / 2: fp16 / bf16 panel for the RTN of
// embed / lm_head, where the reference runs make_*quants in the MODEL dtype
// (quantizer.py:109,195): ATen CPU computes each elementwise op in fp32 and rounds the result
// to the tensor dtype, reductions accumulate in fp32 and round once -- R() below.
#include "gq_common.hpp"

namespace gq {

template <int CTRL>
__device__ __forceinline__ float dpp_f(float old, float src) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL,
                                           0xf, 0xf, false));
}

// sum of the 8 lane partials of an 8-lane group, added in lane order 0 -> 7
// (the ATen order the CPU restatement follows), result valid on EVERY lane of the group.
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

// Group reductions for LPG lanes per group (8: the DPP forms above; 1: the whole group in one lane).
// LPG = 8: element e of a group lives on lane e % 8 (one accumulator per lane).  LPG = 2: the 16-byte chunk c of the
// group lives on lane c % 2, so the even lane owns ATen accumulators 0-3 and the odd lane 4-7.  LPG = 1: everything
// in one lane.  In every mapping local element k feeds local accumulator k % NA (NA = 8 / LPG) in increasing order
// of e, and the eight accumulators are summed 0 -> 7: the same additions in the same order.
template <int LPG, int NA>
__device__ __forceinline__ float group_sum(const float (&acc)[NA]) {
    if constexpr (LPG == 8) {
        return group_sum8(acc[0]);
    } else if constexpr (LPG == 2) {
        // lane pair: the even lane owns accumulators 0-3, the odd lane 4-7.  ((((a0+a1)+a2)+a3)+a4)+..+a7: the even
        // lane's partial travels to the odd lane, which adds its four in order; the total travels back.
        float t = acc[0];
#pragma unroll
        for (int j = 1; j < NA; ++j) t = t + acc[j];
        float u = dpp_f<0xA0>(t, t);  // quad_perm [0,0,2,2]: the even lane's partial on both lanes
#pragma unroll
        for (int j = 0; j < NA; ++j) u = u + acc[j];
        return dpp_f<0xF5>(u, u);     // quad_perm [1,1,3,3]: the odd lane's total on both lanes
    } else {
        float s = acc[0];
#pragma unroll
        for (int j = 1; j < NA; ++j) s = s + acc[j];
        return s;
    }
}
template <int LPG>
__device__ __forceinline__ float group_min(float v) {
    if constexpr (LPG == 8) return group_min8(v);
    else if constexpr (LPG == 2) {
        const float t = dpp_f<0xB1>(v, v);  // quad_perm [1,0,3,2]: the partner lane
        return t < v ? t : v;
    } else return v;
}
template <int LPG>
__device__ __forceinline__ float group_max(float v) {
    if constexpr (LPG == 8) return group_max8(v);
    else if constexpr (LPG == 2) {
        const float t = dpp_f<0xB1>(v, v);
        return t > v ? t : v;
    } else return v;
}
// acc[k % NA] (+)= term, first touch assigns
#define GQ_ACC(acc, k, term)                       \
    do {                                           \
        if ((k) < NA) acc[(k) % NA] = (term);      \
        else acc[(k) % NA] = acc[(k) % NA] + (term); \
    } while (0)

template <int RM>
__device__ __forceinline__ float R(float v) {
    if constexpr (RM == 1) {
        // The empty asm hides "this float is an fpext of a half" from LLVM: otherwise
        // fptrunc(fdiv(fpext a, fpext b)) is narrowed to an fp16 divide, which gfx950 lowers through
        // v_rcp_f32 (approximate) -- measured: 1 group in ~4600 picked another candidate.
        float r = h2f(f2h(v));
        asm volatile("" : "+v"(r));
        return r;
    } else if constexpr (RM == 2) {
        return bf2f(f2bf(v));
    } else {
        return v;
    }
}

struct SearchParams {
    float num[24];  // fp32(rmin + rdelta*i + maxq), evaluated in double on the host
    int nstep;
    int mse_n;       // quant_scale == "mse": candidates of the grid search, int(maxshrink * grid) + 1; 0: absmax
    double mse_den;  // maxshrink * grid
    // Row-stacked matrices (gq_gptq_quantize_stacked: several Linears that share U, one under the other): matrix k is
    // rows [row_end[k-1], row_end[k]), every boundary a multiple of 64.  The one cross-row dependence of the path -- the
    // panel-wide `continue` of quant_utils.py:250-252 -- is per MATRIX: one pair of panel words per stacked matrix.
    int nstack;      // <= 1: one matrix
    int row_end[GQ_MAX_STACK];
};

__device__ __forceinline__ int stack_of(const SearchParams& sp, int64_t row) {
    int k = 0;
    for (int i = 0; i + 1 < sp.nstack; ++i) k += row >= sp.row_end[i] ? 1 : 0;
    return k;
}

// quant_utils.py:199-274 for one group spread over LPG lanes; NS = G/LPG values per lane.
// `skip`: iterations the WHOLE panel skips (bit i); `valid`: bit i set when this group has D > eps in iteration i;
// `accepted`: bit i set when this group took the candidate of iteration i -- what panel_fixup_kernel needs to follow
// the panel-wide `if not valid.any(): continue` of :250-252.
template <int NS, int BITS, int RM, int LPG>
__device__ __forceinline__ void k_search(const float (&x)[NS], const SearchParams& sp, float& scale_out,
                                         float& zero_out, unsigned skip, unsigned& valid, unsigned& accepted) {
    constexpr float maxq = (float)((1 << BITS) - 1);
    constexpr float G = (float)(NS * LPG);
    constexpr int NA = 8 / LPG;
    const float eps = R<RM>(1e-9f);  // quant_utils.py:69; rounds to 0 in fp16
    float w[NS];
    float a0[NA], a1[NA], a2[NA];
    // :203-205
#pragma unroll
    for (int k = 0; k < NS; ++k) GQ_ACC(a0, k, R<RM>(x[k] * x[k]));
    float sum_x2 = R<RM>(group_sum<LPG, NA>(a0));
    float av_x = R<RM>(sqrtf(R<RM>(sum_x2 / G)));  // IEEE sqrt (MKL vsSqrt note)
#pragma unroll
    for (int k = 0; k < NS; ++k) w[k] = R<RM>(av_x + fabsf(x[k]));
    // :208-211
    float mn = x[0], mx = x[0];
#pragma unroll
    for (int k = 1; k < NS; ++k) {
        mn = x[k] < mn ? x[k] : mn;
        mx = x[k] > mx ? x[k] : mx;
    }
    mn = group_min<LPG>(mn);
    mx = group_max<LPG>(mx);
    mn = mn < 0.0f ? mn : 0.0f;
    float x_min = mn;
    const float x_max = mx;
    const bool is_const = (x_max == x_min);
    // :214-215
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        GQ_ACC(a0, k, w[k]);
        GQ_ACC(a1, k, R<RM>(w[k] * x[k]));
    }
    const float sum_w = R<RM>(group_sum<LPG, NA>(a0));
    const float sum_x = R<RM>(group_sum<LPG, NA>(a1));
    // :218-232
    float sc = R<RM>(R<RM>(x_max - x_min) / maxq);
    if (is_const) sc = 0.0f;
    const float isc = R<RM>(1.0f / (sc < eps ? eps : sc));
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        float q = 0.0f;
        if (!is_const) q = clampf(rintf(R<RM>(R<RM>(x[k] - x_min) * isc)), 0.0f, maxq);
        float diff = R<RM>(R<RM>(R<RM>(sc * q) + x_min) - x[k]);
        GQ_ACC(a0, k, R<RM>(w[k] * R<RM>(diff * diff)));
    }
    float best_err = R<RM>(group_sum<LPG, NA>(a0));
    float best_scale = sc;

    if (sp.nstep >= 1) {  // :235-237
        for (int i = 0; i <= sp.nstep; ++i) {  // :240
            if ((skip >> i) & 1u) continue;     // :250-252, decided for the whole panel
            // :241 scalar/tensor == reciprocal()*scalar; x_min is the aliased best_min (:228,:270)
            float den = R<RM>(x_max - x_min);
            den = den < eps ? eps : den;
            const float cand_iscale = R<RM>(R<RM>(1.0f / den) * sp.num[i]);
            float L[NS];
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                float q = 0.0f;  // :243 const groups
                if (!is_const) q = clampf(rintf(R<RM>(R<RM>(x[k] - x_min) * cand_iscale)), 0.0f, maxq);  // :242
                float q2;  // :246 new_q**2 stays uint8: wraps for Q5_K, exact in fp32 below 16
                if constexpr (BITS <= 4) {
                    q2 = q * q;
                } else {
                    int qi = (int)q;
                    q2 = (float)((qi * qi) & 255);
                }
                L[k] = q;
                GQ_ACC(a0, k, R<RM>(w[k] * q));
                GQ_ACC(a1, k, R<RM>(w[k] * q2));
                GQ_ACC(a2, k, R<RM>(R<RM>(w[k] * x[k]) * q));
            }
            const float sum_l = R<RM>(group_sum<LPG, NA>(a0));
            const float sum_l2 = R<RM>(group_sum<LPG, NA>(a1));
            const float sum_xl = R<RM>(group_sum<LPG, NA>(a2));
            const float D = R<RM>(R<RM>(sum_w * sum_l2) - R<RM>(sum_l * sum_l));                           // :249
            valid |= (D > eps ? 1u : 0u) << i;                                                              // :250
            float this_scale = R<RM>(R<RM>(R<RM>(sum_w * sum_xl) - R<RM>(sum_x * sum_l)) / D);             // :254
            float this_min = R<RM>(R<RM>(R<RM>(sum_l2 * sum_x) - R<RM>(sum_l * sum_xl)) / D);              // :255
            if (this_min > 0.0f) {                                                                          // :257-260
                this_scale = R<RM>(sum_xl / (sum_l2 < eps ? eps : sum_l2));
                this_min = 0.0f;
            }
#pragma unroll
            for (int k = 0; k < NS; ++k) {  // :262-264
                float diff = R<RM>(R<RM>(R<RM>(this_scale * L[k]) + this_min) - x[k]);
                GQ_ACC(a0, k, R<RM>(w[k] * R<RM>(diff * diff)));
            }
            const float cand_err = R<RM>(group_sum<LPG, NA>(a0));
            // :250-252: a group with D <= eps still goes through the formulas as long as SOME group of the panel is
            // valid in this iteration (NaN / inf candidates lose the comparison); when NONE is, the reference skips
            // the iteration for everyone -- detected through `valid`, redone by panel_fixup_kernel.
            if (cand_err < best_err) {  // :266-271 (NaN compares false)
                accepted |= 1u << i;
                best_err = cand_err;
                best_scale = this_scale;
                x_min = this_min;
            }
        }
    }
    scale_out = best_scale;
    zero_out = -x_min;  // :273
}

// quant_utils.py:147-197: absmax, and with sp.mse_n > 0 the "mse" grid search of :164-191 -- verbatim, i.e. with the
// .round() that lands on the SCALE (:180: for scales <= 0.5 the divisor is 0, every candidate gives q_int = 0 and the
// first one, the absmax scale, is kept) and with the un-rounded q_int.
template <int NS, int BITS, int RM, int LPG>
__device__ __forceinline__ void absmax_search(const float (&x)[NS], const SearchParams& sp, float& scale_out,
                                              float& zero_out) {
    constexpr float maxq = (float)((1 << BITS) - 1);
    float mn = x[0], mx = x[0];
#pragma unroll
    for (int k = 1; k < NS; ++k) {
        mn = x[k] < mn ? x[k] : mn;
        mx = x[k] > mx ? x[k] : mx;
    }
    mn = group_min<LPG>(mn);
    mx = group_max<LPG>(mx);
    float a = fabsf(mn);
    mx = a > mx ? a : mx;     // :153
    if (mn < 0.0f) mn = -mx;  // :154-156
    if (mn == mx) {           // :157-159
        mn = -1.0f;
        mx = 1.0f;
    }
    scale_out = R<RM>(R<RM>(mx - mn) / maxq);  // :161
    zero_out = 0.0f;                           // :195
    if (sp.mse_n > 0) {
        constexpr int NA = 8 / LPG;
        const float zq = R<RM>((maxq + 1.0f) / 2.0f);  // :162
        const float eps = R<RM>(1e-9f);
        float amax = fabsf(mn);
        amax = mx > amax ? mx : amax;  // :171
        float min_loss = __builtin_inff(), best = 0.0f;
        for (int i = 0; i < sp.mse_n; ++i) {
            const float alpha = R<RM>((float)(1.0 - (double)i / sp.mse_den));  // :170
            const float cand = R<RM>(amax * alpha);
            const float xmax1 = mx < cand ? mx : cand;       // :173
            const float xmin1 = mn > -cand ? mn : -cand;     // :174
            const float scale1 = R<RM>(R<RM>(xmax1 - xmin1) / maxq);  // :176
            float c = scale1 < eps ? eps : scale1;
            if (scale1 != scale1) c = scale1;
            const float dv = rintf(c);  // :180
            float a0[NA];
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const float q = clampf(R<RM>(R<RM>(x[k] - zq) / dv), 0.0f, maxq);  // :179-181 (NaN stays NaN)
                const float y = R<RM>(R<RM>(q * scale1) + zq);                      // :182
                const float df = R<RM>(y - x[k]);
                GQ_ACC(a0, k, R<RM>(df * df));                                      // :183
            }
            const float loss = R<RM>(group_sum<LPG, NA>(a0));
            if (loss < min_loss) {  // :185-189
                min_loss = loss;
                best = scale1;
            }
        }
        scale_out = best;  // :190
    }
}

template <int RM>
__device__ __forceinline__ float load_x(const void* x, int64_t idx) {
    if constexpr (RM == 0) return reinterpret_cast<const float*>(x)[idx];
    else if constexpr (RM == 1) return h2f(reinterpret_cast<const uint16_t*>(x)[idx]);
    else return bf2f(reinterpret_cast<const uint16_t*>(x)[idx]);
}

// OR of the lanes' per-iteration bits into the panel words: [0] valid, [1] accepted (one atomic per wave and word,
// skipped when nothing is new)
__device__ __forceinline__ void publish_valid(unsigned* panel, unsigned v, unsigned a) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        v |= (unsigned)__shfl_xor((int)v, o);
        a |= (unsigned)__shfl_xor((int)a, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if ((__hip_atomic_load(panel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & v) != v) atomicOr(panel, v);
        if ((__hip_atomic_load(panel + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & a) != a) atomicOr(panel + 1, a);
    }
}

// One workgroup = 256 threads = 256/(NG*8) row-panels.  NG = 256/G groups per row.
template <int GSZ, int BITS, bool KSEARCH, bool SIGNED, int SMQ, int RM>
__global__ __launch_bounds__(256) void scale_search_kernel(
    const void* __restrict__ x, int64_t rows, int64_t ld, SearchParams sp,
    uint16_t* __restrict__ d, int64_t d_stride, uint8_t* __restrict__ s, int64_t s_ld,
    uint16_t* __restrict__ dmin, int64_t dmin_stride, uint8_t* __restrict__ m, int64_t m_ld,
    float* __restrict__ gs_out, float* __restrict__ gz_out, unsigned* panel_valid) {
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
    const int64_t base = (live ? row : 0) * ld + g * GSZ + l8;
    float xv[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) xv[k] = load_x<RM>(x, base + k * 8);

    float gscale, gzero;
    if constexpr (KSEARCH) {
        unsigned valid = 0, accepted = 0;
        k_search<NS, BITS, RM, 8>(xv, sp, gscale, gzero, 0u, valid, accepted);
        if (panel_valid) publish_valid(panel_valid + 2 * stack_of(sp, live ? row : rows - 1), live ? valid : 0u, live ? accepted : 0u);
    } else {
        absmax_search<NS, BITS, RM, 8>(xv, sp, gscale, gzero);
    }
    if (l8 == 0) {
        sh_scale[row_l][g] = gscale;
        sh_zero[row_l][g] = gzero;
        if (gs_out && live) {  // make_k_quants / make_quants outputs (gq_group_search)
            gs_out[row * NG + g] = gscale;
            gz_out[row * NG + g] = gzero;
        }
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
        float inv_scale = max_scale > 0.0f ? R<RM>(R<RM>(1.0f / max_scale) * smq) : 0.0f;  // :128
        float inv_zero = max_zero > 0.0f ? R<RM>(R<RM>(1.0f / max_zero) * smq) : 0.0f;     // :129
        float a = clampf(rintf(R<RM>(inv_scale * gscale)), 0.0f, smq);                     // :132-143
        float b = clampf(rintf(R<RM>(inv_zero * gzero)), 0.0f, smq);
        s[row * s_ld + g] = SIGNED ? (uint8_t)(int8_t)a : (uint8_t)a;
        m[row * m_ld + g] = SIGNED ? (uint8_t)(int8_t)b : (uint8_t)b;
        if (g == 0) {
            d[row * d_stride] = f2h(R<RM>(max_scale / smq));       // :124 (+ .to(float16))
            dmin[row * dmin_stride] = f2h(R<RM>(max_zero / smq));  // :125
        }
    }
}

// Lane-per-group variant: one lane owns a whole group (its GSZ values in registers), one wave64 =
// 64/NG row-panels.  Same arithmetic in the same order as the kernel above (see group_sum), but no
// cross-lane reductions and the per-group scalar algebra is done once instead of on 8 lanes: about
// half the VALU work per group.  Rows are read with 16-B loads (the launcher checks alignment).
template <int GSZ, int BITS, bool KSEARCH, bool SIGNED, int SMQ, int RM, int LPG>
__global__ __launch_bounds__(64) void scale_search_lane_kernel(
    const void* __restrict__ x, int64_t rows, int64_t ld, SearchParams sp,
    uint16_t* __restrict__ d, int64_t d_stride, uint8_t* __restrict__ s, int64_t s_ld,
    uint16_t* __restrict__ dmin, int64_t dmin_stride, uint8_t* __restrict__ m, int64_t m_ld,
    float* __restrict__ gs_out, float* __restrict__ gz_out, unsigned* panel_valid) {
    static_assert(LPG == 1 || LPG == 2, "one lane or a lane pair per group");
    constexpr int NG = 256 / GSZ;        // groups per row
    constexpr int LPR = NG * LPG;        // lanes per row
    constexpr int RPW = 64 / LPR;        // rows per wave
    constexpr int NS = GSZ / LPG;        // values per lane
    const int lane = threadIdx.x;
    const int row_l = lane / LPR;
    const int g = (lane % LPR) / LPG;
    const int h = lane % LPG;            // which 16-byte chunks of the group: c % LPG == h
    const int64_t row = (int64_t)blockIdx.x * RPW + row_l;
    const bool live = row < rows;
    const int64_t base = (live ? row : 0) * ld + g * GSZ;
    float xv[NS];
    if constexpr (RM == 0) {
        const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(x) + base);
#pragma unroll
        for (int k = 0; k < NS / 4; ++k) {
            float4 v = p[k * LPG + h];
            xv[4 * k] = v.x; xv[4 * k + 1] = v.y; xv[4 * k + 2] = v.z; xv[4 * k + 3] = v.w;
        }
    } else {
        const uint2* p = reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(x) + base);
#pragma unroll
        for (int k = 0; k < NS / 4; ++k) {  // chunks of four 16-bit values
            uint2 v = p[k * LPG + h];
            const uint32_t u[2] = {v.x, v.y};
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                uint16_t lo = (uint16_t)(u[t] & 0xffffu), hi = (uint16_t)(u[t] >> 16);
                xv[4 * k + 2 * t] = RM == 1 ? h2f(lo) : bf2f(lo);
                xv[4 * k + 2 * t + 1] = RM == 1 ? h2f(hi) : bf2f(hi);
            }
        }
    }

    float gscale, gzero;
    if constexpr (KSEARCH) {
        unsigned valid = 0, accepted = 0;
        k_search<NS, BITS, RM, LPG>(xv, sp, gscale, gzero, 0u, valid, accepted);
        if (panel_valid) publish_valid(panel_valid + 2 * stack_of(sp, live ? row : rows - 1), live ? valid : 0u, live ? accepted : 0u);
    } else {
        absmax_search<NS, BITS, RM, LPG>(xv, sp, gscale, gzero);
    }
    if (gs_out && live && h == 0) {  // make_k_quants / make_quants outputs (gq_group_search)
        gs_out[row * NG + g] = gscale;
        gz_out[row * NG + g] = gzero;
    }
    // quant_utils.py:121-143: row maxima over the lanes of the row
    float max_scale = gscale, max_zero = gzero;
#pragma unroll
    for (int o = LPG; o < LPR; o <<= 1) {
        float a = __shfl_xor(max_scale, o), b = __shfl_xor(max_zero, o);
        max_scale = a > max_scale ? a : max_scale;
        max_zero = b > max_zero ? b : max_zero;
    }
    if (live && h == 0) {
        constexpr float smq = (float)SMQ;
        float inv_scale = max_scale > 0.0f ? R<RM>(R<RM>(1.0f / max_scale) * smq) : 0.0f;  // :128
        float inv_zero = max_zero > 0.0f ? R<RM>(R<RM>(1.0f / max_zero) * smq) : 0.0f;     // :129
        float a = clampf(rintf(R<RM>(inv_scale * gscale)), 0.0f, smq);                     // :132-143
        float b = clampf(rintf(R<RM>(inv_zero * gzero)), 0.0f, smq);
        s[row * s_ld + g] = SIGNED ? (uint8_t)(int8_t)a : (uint8_t)a;
        m[row * m_ld + g] = SIGNED ? (uint8_t)(int8_t)b : (uint8_t)b;
        if (g == 0) {
            d[row * d_stride] = f2h(R<RM>(max_scale / smq));       // :124 (+ .to(float16))
            dmin[row * dmin_stride] = f2h(R<RM>(max_zero / smq));  // :125
        }
    }
}

// quant_utils.py:250-252: `if not valid.any(): continue` looks at ALL groups of the [rows, 256] panel.  The search
// kernels above run every iteration and record, per iteration, whether any group was valid.  This one-workgroup
// kernel follows every search launch: if every iteration had a valid group (any panel with an entry above ~1e-6) it
// returns at once.  Otherwise the reference skipped iterations the kernels ran: with S = the set of skipped
// iterations (initially empty), the first iteration outside S without a valid group is one the reference skips --
// everything before it ran on the right state -- so it joins S; if some group had TAKEN that iteration's candidate
// the whole panel is searched again with S (its state differs from there on), else nothing changes downstream.
// Repeated until no such iteration is left (at most nstep + 1 searches; a panel of ~1e-8 values or of constant
// groups, never a weight matrix -- Q5_K panels take the no-search branch every time).  One lane per group, plain loads: the slow path only has to be right.  Leaves the panel word at 0.
template <int GSZ, int BITS, bool SIGNED, int SMQ, int RM>
__global__ __launch_bounds__(256) void panel_fixup_kernel(
    const void* x, int64_t rows, int64_t ld, SearchParams sp,
    uint16_t* d, int64_t d_stride, uint8_t* s, int64_t s_ld,
    uint16_t* dmin, int64_t dmin_stride, uint8_t* m, int64_t m_ld,
    float* gs_out, float* gz_out, unsigned* panel_valid) {
    constexpr int NG = 256 / GSZ;
    constexpr int RPB = 256 / NG;  // rows per pass of the workgroup
    __shared__ unsigned sh_valid, sh_acc;
    // word GQ_PANEL_RESEARCH of the call chain's panel block: how many times a panel was searched AGAIN (sticky: the chain's
    // owner zeroes and reads it).  A row slice of a row-split matrix decides `valid.any()` over its own rows; as long as no
    // slice ever searched again, every slice's results equal the whole matrix's -- the host checks this word
    unsigned* const researches = panel_valid + GQ_PANEL_RESEARCH;
    if (sp.nstack > 1) {  // workgroup k = stacked matrix k: its rows, its panel words
        const int64_t r0 = blockIdx.x ? sp.row_end[blockIdx.x - 1] : 0;
        rows = sp.row_end[blockIdx.x] - r0;
        x = RM == 0 ? (const void*)(reinterpret_cast<const float*>(x) + r0 * ld)
                    : (const void*)(reinterpret_cast<const uint16_t*>(x) + r0 * ld);
        d += r0 * d_stride; dmin += r0 * dmin_stride; s += r0 * s_ld; m += r0 * m_ld;
        if (gs_out) { gs_out += r0 * NG; gz_out += r0 * NG; }
        panel_valid += 2 * blockIdx.x;
    }
    const unsigned full = (2u << sp.nstep) - 1u;
    unsigned V = panel_valid[0] & full, A = panel_valid[1] & full;
    unsigned skip = 0;
    const int tid = threadIdx.x, g = tid % NG, row_l = tid / NG;
    while (true) {
        const unsigned missing = full & ~(V | skip);
        if (missing == 0) break;              // uniform: V, A and skip are the same in every thread
        const unsigned i0 = missing & (0u - missing);  // the lowest iteration nobody was valid in: the reference skips it
        skip |= i0;
        // nobody TOOK its candidate either (the usual case: Q5_K's uint8-wrapped squares make D negative for every
        // group of ordinary weights, and such candidates never win): skipping it changes no state, the bits of the
        // later iterations stand
        if ((A & i0) == 0) continue;
        if (tid == 0) {
            sh_valid = 0;
            sh_acc = 0;
            atomicAdd(researches, 1u);
        }
        __syncthreads();
        for (int64_t r0 = 0; r0 < rows; r0 += RPB) {
            const int64_t row = r0 + row_l;
            const bool live = row < rows;
            const int64_t base = (live ? row : 0) * ld + g * GSZ;
            float xv[GSZ];
#pragma unroll
            for (int k = 0; k < GSZ; ++k) xv[k] = load_x<RM>(x, base + k);
            float gscale, gzero;
            unsigned valid = 0, accepted = 0;
            k_search<GSZ, BITS, RM, 1>(xv, sp, gscale, gzero, skip, valid, accepted);
            if (live && valid) atomicOr(&sh_valid, valid);
            if (live && accepted) atomicOr(&sh_acc, accepted);
            if (gs_out && live) {
                gs_out[row * NG + g] = gscale;
                gz_out[row * NG + g] = gzero;
            }
            float max_scale = gscale, max_zero = gzero;  // quant_utils.py:121-143, as in the kernels above
#pragma unroll
            for (int o = 1; o < NG; o <<= 1) {
                float a = __shfl_xor(max_scale, o), b = __shfl_xor(max_zero, o);
                max_scale = a > max_scale ? a : max_scale;
                max_zero = b > max_zero ? b : max_zero;
            }
            if (live) {
                constexpr float smq = (float)SMQ;
                float inv_scale = max_scale > 0.0f ? R<RM>(R<RM>(1.0f / max_scale) * smq) : 0.0f;
                float inv_zero = max_zero > 0.0f ? R<RM>(R<RM>(1.0f / max_zero) * smq) : 0.0f;
                float a = clampf(rintf(R<RM>(inv_scale * gscale)), 0.0f, smq);
                float b = clampf(rintf(R<RM>(inv_zero * gzero)), 0.0f, smq);
                s[row * s_ld + g] = SIGNED ? (uint8_t)(int8_t)a : (uint8_t)a;
                m[row * m_ld + g] = SIGNED ? (uint8_t)(int8_t)b : (uint8_t)b;
                if (g == 0) {
                    d[row * d_stride] = f2h(R<RM>(max_scale / smq));
                    dmin[row * dmin_stride] = f2h(R<RM>(max_zero / smq));
                }
            }
        }
        __syncthreads();
        V = sh_valid & full;
        A = sh_acc & full;
        __syncthreads();
    }
    if (tid == 0) {
        panel_valid[0] = 0;
        panel_valid[1] = 0;
    }
}

template <int RM>
static int launch_ss(const void* x, int64_t rows, int64_t ld, int q_type, const gq_search_t* p, uint16_t* d,
                     int64_t d_stride, uint8_t* s, int64_t s_ld, uint16_t* dmin, int64_t dmin_stride, uint8_t* m,
                     int64_t m_ld, hipStream_t st, float* gs_out = nullptr, float* gz_out = nullptr,
                     unsigned* panel = nullptr, const int64_t* row_ends = nullptr, int nstack = 1) {
    // `panel`: one zeroed device word per call chain (left at zero again); nullptr: taken from the stream's pool
    TypeInfo ti;
    if (!type_info(q_type, ti)) GQ_FAIL(GQ_E_BAD_TYPE, "gq_scale_search: unknown q_type %d", q_type);
    if (rows <= 0 || ld < 256) GQ_FAIL(GQ_E_BAD_SHAPE, "gq_scale_search: rows=%ld ld=%ld", (long)rows, (long)ld);
    void* own = nullptr;
    if (nstack > 1) {
        if (nstack > GQ_MAX_STACK || !row_ends) GQ_FAIL(GQ_E_UNSUPPORTED, "gq_scale_search: %d stacked matrices (at most %d)", nstack, GQ_MAX_STACK);
        for (int k = 0; k < nstack; ++k)
            if (row_ends[k] % 64 || row_ends[k] <= (k ? row_ends[k - 1] : 0) || row_ends[k] > rows || row_ends[k] > INT32_MAX)
                GQ_FAIL(GQ_E_BAD_SHAPE, "gq_scale_search: stacked row boundary %ld (ascending multiples of 64 up to rows=%ld)", (long)row_ends[k], (long)rows);
        if (row_ends[nstack - 1] != rows) GQ_FAIL(GQ_E_BAD_SHAPE, "gq_scale_search: the last stacked matrix ends at %ld, rows=%ld", (long)row_ends[nstack - 1], (long)rows);
    }
    if (ti.k_search && !panel && (!p || p->nstep >= 1)) {
        GQ_HIP(hipMallocAsync(&own, 256, st));
        GQ_HIP(hipMemsetAsync(own, 0, 256, st));
        panel = reinterpret_cast<unsigned*>(own);
    }
    SearchParams sp;
    sp.nstack = nstack > 1 ? nstack : 1;
    for (int k = 0; k < GQ_MAX_STACK; ++k) sp.row_end[k] = (nstack > 1 && k < nstack) ? (int)row_ends[k] : 0;
    sp.nstep = p ? p->nstep : 20;
    if (sp.nstep > 23) GQ_FAIL(GQ_E_UNSUPPORTED, "gq_scale_search: nstep=%d > 23", sp.nstep);
    const double rmin = p ? p->rmin : -1.0, rdelta = p ? p->rdelta : 0.1;
    const double maxq = (double)((1 << ti.bits) - 1);
    for (int i = 0; i < 24; ++i) sp.num[i] = (float)(rmin + rdelta * (double)i + maxq);
    sp.mse_n = 0;
    sp.mse_den = 1.0;
    if (p && p->quant_scale == 1 && !ti.k_search) {  // make_k_quants ignores quant_scale
        sp.mse_den = p->maxshrink * (double)p->grid;
        sp.mse_n = (int)sp.mse_den + 1;
        if (!(sp.mse_den > 0.0) || sp.mse_n > 100000)
            GQ_FAIL(GQ_E_UNSUPPORTED, "gq_scale_search: quant_scale=mse with grid=%d maxshrink=%g", p->grid, p->maxshrink);
    } else if (p && p->quant_scale != 0 && p->quant_scale != 1) {
        GQ_FAIL(GQ_E_UNSUPPORTED, "gq_scale_search: quant_scale=%d (0: absmax, 1: mse)", p->quant_scale);
    }
    // Three mappings with identical arithmetic (profiles/ss_probe.py): one lane per group does half the VALU work
    // of the 8-lane kernel but a wave takes ~35 us whatever the size -- used from one wave per SIMD up
    // (rows * groups >= 65536); a lane PAIR per group halves that latency at +12 % work -- the middle range, where the
    // launch is latency-bound (a 4096-row Q4_K panel: 1024 waves, one per SIMD: 23 us against 36 / 31 us for the other
    // two; rows * groups >= 16384); 8 lanes per group for small
    // panels and rows that cannot be read with 16-byte (8-byte for 16-bit inputs) loads.
    // option ss_wide = 1 / 0 / 2 forces the 8-lane / 1-lane / 2-lane kernel (A/B measurements, parity test).
    const size_t esz = RM == 0 ? 4 : 2;
    const bool aligned = (reinterpret_cast<uintptr_t>(x) % 16 == 0) && ((size_t)ld * esz) % 16 == 0;
    const int64_t wide = opt(OPT_ss_wide);
    const int64_t ngroups = rows * (256 / ti.group);
    int lpg = !aligned ? 8 : (ngroups >= 65536 ? 1 : (ngroups >= 16384 ? 2 : 8));
    if (wide == 1) lpg = 8;
    if (wide == 0 && aligned) lpg = 1;
    if (wide == 2 && aligned) lpg = 2;
    const int rpw = lpg == 8 ? (ti.group == 32 ? 4 : 2) : 64 / ((256 / ti.group) * lpg);
    dim3 grid((unsigned)((rows + rpw - 1) / rpw)), block(lpg == 8 ? 256 : 64);
    ProfScope ps(PT_SCALE_SEARCH, st);
#define GQ_SS(G, B, K, S, Q)                                                                                       \
    do {                                                                                                           \
        if (lpg == 1)                                                                                              \
            hipLaunchKernelGGL((scale_search_lane_kernel<G, B, K, S, Q, RM, 1>), grid, block, 0, st, x, rows, ld,  \
                               sp, d, d_stride, s, s_ld, dmin, dmin_stride, m, m_ld, gs_out, gz_out, panel);       \
        else if (lpg == 2)                                                                                         \
            hipLaunchKernelGGL((scale_search_lane_kernel<G, B, K, S, Q, RM, 2>), grid, block, 0, st, x, rows, ld,  \
                               sp, d, d_stride, s, s_ld, dmin, dmin_stride, m, m_ld, gs_out, gz_out, panel);       \
        else                                                                                                       \
            hipLaunchKernelGGL((scale_search_kernel<G, B, K, S, Q, RM>), grid, block, 0, st, x, rows, ld, sp, d,   \
                               d_stride, s, s_ld, dmin, dmin_stride, m, m_ld, gs_out, gz_out, panel);              \
        if (K && panel && sp.nstep >= 1)                                                                           \
            hipLaunchKernelGGL((panel_fixup_kernel<G, B, S, Q, RM>), dim3((unsigned)sp.nstack), dim3(256), 0, st, x, rows, ld, sp, \
                               d, d_stride, s, s_ld, dmin, dmin_stride, m, m_ld, gs_out, gz_out, panel);           \
    } while (0)
    switch (q_type) {
    case GQ_Q2_K: GQ_SS(16, 2, true, false, 15); break;
    case GQ_Q3_K: GQ_SS(16, 3, false, true, 31); break;
    case GQ_Q4_K: GQ_SS(32, 4, true, false, 63); break;
    case GQ_Q5_K: GQ_SS(32, 5, true, false, 63); break;
    case GQ_Q6_K: GQ_SS(16, 6, false, true, 63); break;
    }
#undef GQ_SS
    GQ_LAUNCH_CHECK();
    if (own) GQ_HIP(hipFreeAsync(own, st));
    return GQ_OK;
}

int launch_scale_search(const float* x, int64_t rows, int64_t ld, int q_type, const gq_search_t* p,
                        uint16_t* d, int64_t d_stride, uint8_t* s, int64_t s_ld, uint16_t* dmin,
                        int64_t dmin_stride, uint8_t* m, int64_t m_ld, hipStream_t st, unsigned* panel,
                        const int64_t* row_ends, int nstack) {
    return launch_ss<0>(x, rows, ld, q_type, p, d, d_stride, s, s_ld, dmin, dmin_stride, m, m_ld, st, nullptr, nullptr,
                        panel, row_ends, nstack);
}

// make_k_quants / make_quants outputs (per-group fp32 scale and zero) of one [rows,256] panel in
// x_dtype, next to the super-group outputs: the reference functions quant_utils.py:147-274 themselves.
int group_search(const void* x, int x_dtype, int64_t rows, int64_t ld, int q_type, const gq_search_t* p, float* gs,
                 float* gz, uint16_t* d, uint8_t* s, uint16_t* dmin, uint8_t* m, hipStream_t st) {
    TypeInfo ti;
    if (!type_info(q_type, ti)) GQ_FAIL(GQ_E_BAD_TYPE, "gq_group_search: unknown q_type %d", q_type);
    if (!x || !gs || !gz || !d || !s || !dmin || !m) GQ_FAIL(GQ_E_NULL, "gq_group_search: null pointer");
    const int ng = 256 / ti.group;
    switch (x_dtype) {
    case GQ_F32: return launch_ss<0>(x, rows, ld, q_type, p, d, 1, s, ng, dmin, 1, m, ng, st, gs, gz);
    case GQ_F16: return launch_ss<1>(x, rows, ld, q_type, p, d, 1, s, ng, dmin, 1, m, ng, st, gs, gz);
    case GQ_BF16: return launch_ss<2>(x, rows, ld, q_type, p, d, 1, s, ng, dmin, 1, m, ng, st, gs, gz);
    default: GQ_FAIL(GQ_E_BAD_TYPE, "gq_group_search: unknown x_dtype %d", x_dtype);
    }
}

// RTN scale search in the model dtype (quantizer.py:300-310 over every super-group of the
// unmodified fp16 / bf16 weight).
int launch_rtn_scale_search(const void* W, int w_dtype, int64_t R_, int64_t C, int q_type, const gq_search_t* p,
                            uint16_t* d, uint8_t* s, uint16_t* dmin, uint8_t* m, hipStream_t st) {
    TypeInfo ti;
    if (!type_info(q_type, ti)) GQ_FAIL(GQ_E_BAD_TYPE, "gq_rtn_quantize: unknown q_type %d", q_type);
    if (w_dtype != GQ_F16 && w_dtype != GQ_BF16) GQ_FAIL(GQ_E_BAD_TYPE, "gq_rtn_quantize: unknown w_dtype %d", w_dtype);
    const int64_t ng = C / ti.group, nsg = C / 256;
    const int gps = 256 / ti.group;
    const uint16_t* Wh = reinterpret_cast<const uint16_t*>(W);
    for (int64_t c = 0; c < C; c += 256) {
        int rc = w_dtype == GQ_F16
                     ? launch_ss<1>(Wh + c, R_, C, q_type, p, d + c / 256, nsg, s + (c / 256) * gps, ng, dmin + c / 256,
                                    nsg, m + (c / 256) * gps, ng, st)
                     : launch_ss<2>(Wh + c, R_, C, q_type, p, d + c / 256, nsg, s + (c / 256) * gps, ng, dmin + c / 256,
                                    nsg, m + (c / 256) * gps, ng, st);
        if (rc) return rc;
    }
    return GQ_OK;
}

}  // namespace gq
