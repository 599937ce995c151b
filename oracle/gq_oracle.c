/*
 * gq_oracle.c -- CPU ORACLE (test infrastructure, not product code).
 * See gq_oracle.h for the contract.  Build: oracle/Makefile
 *   gcc -O2 -ffp-contract=off -mfma -fopenmp ...
 * -ffp-contract=off is REQUIRED: every a*b+c below is two roundings unless it is
 * spelled fmaf().  File:line citations are into /root/reference/quant/gptq/src/.
 */
#include "gq_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ types */

int gqo_type_info(int q_type, gqo_type_info_t* o) {
    /* quant_utils.py:19-26; type sizes from gguf.constants.GGML_QUANT_SIZES */
    switch (q_type) {
    case GQO_Q2_K: *o = (gqo_type_info_t){2, 0, 3, 15, 16, 0, 1, 84}; return 0;
    case GQO_Q3_K: *o = (gqo_type_info_t){3, -4, 3, 31, 16, 1, 0, 110}; return 0;
    case GQO_Q4_K: *o = (gqo_type_info_t){4, 0, 15, 63, 32, 0, 1, 144}; return 0;
    case GQO_Q5_K: *o = (gqo_type_info_t){5, 0, 31, 63, 32, 0, 1, 176}; return 0;
    case GQO_Q6_K: *o = (gqo_type_info_t){6, -32, 31, 63, 16, 1, 0, 210}; return 0;
    default: return -1;
    }
}

/* ------------------------------------------------------------- fp16/bf16 */

uint16_t gqo_f32_to_f16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t absx = x & 0x7fffffffu;
    if (absx >= 0x7f800000u) { /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | ((absx > 0x7f800000u) ? 0x200u | ((absx >> 13) & 0x3ffu) : 0));
    }
    if (absx >= 0x477ff000u) { /* >= 65520 rounds to inf */
        return (uint16_t)(sign | 0x7c00u);
    }
    if (absx < 0x33000001u) { /* < 2^-25 (or == 2^-25 exactly: ties to even -> 0) */
        return (uint16_t)sign;
    }
    int32_t e = (int32_t)(absx >> 23) - 127;
    uint32_t mant = (absx & 0x7fffffu) | 0x800000u;
    if (e < -14) { /* subnormal half */
        int shift = (-14 - e) + 13; /* bits to drop from the 24-bit mantissa */
        uint32_t h = mant >> shift;
        uint32_t rem = mant & ((1u << shift) - 1);
        uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1))) h++;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((uint32_t)(e + 15) << 10) | ((mant >> 13) & 0x3ffu);
    uint32_t rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) h++;
    return (uint16_t)(sign | h);
}

float gqo_f16_to_f32(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1f;
    uint32_t m = h & 0x3ff;
    uint32_t x;
    if (e == 0) {
        if (m == 0) {
            x = sign;
        } else {
            int sh = 0;
            while (!(m & 0x400)) { m <<= 1; sh++; }
            m &= 0x3ff;
            x = sign | ((uint32_t)(127 - 15 - sh + 1) << 23) | (m << 13);
        }
    } else if (e == 31) {
        x = sign | 0x7f800000u | (m << 13);
    } else {
        x = sign | ((e + 112) << 23) | (m << 13);
    }
    float f;
    memcpy(&f, &x, 4);
    return f;
}

uint16_t gqo_f32_to_bf16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    if ((x & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((x >> 16) | 0x40);
    uint32_t lsb = (x >> 16) & 1;
    x += 0x7fffu + lsb;
    return (uint16_t)(x >> 16);
}

float gqo_bf16_to_f32(uint16_t h) {
    uint32_t x = (uint32_t)h << 16;
    float f;
    memcpy(&f, &x, 4);
    return f;
}

/* ------------------------------------------------------- ATen sum order */

/* ATen CPU inner-dim fp32 sum over n contiguous values, n in {16, 32}: eight
   lanes; lane l accumulates v[l], v[l+8], ... in order; the eight lane sums are
   then added 0 -> 7.  Matches torch 2.10 CPU (AVX512 build) on 2000/2000 random
   rows for n=16 and n=32 -- see tests/test_oracle_golden.py::test_aten_sum. */
float gqo_aten_sum(const float* v, int n) {
    float lane[8];
    for (int l = 0; l < 8; ++l) lane[l] = v[l];
    for (int j = 8; j < n; j += 8)
        for (int l = 0; l < 8; ++l) lane[l] = lane[l] + v[j + l];
    float s = lane[0];
    for (int l = 1; l < 8; ++l) s = s + lane[l];
    return s;
}

/* Reduced-precision emulation (reference quantizer.py:109,195: embed/lm_head RTN runs make_*quants in
   the MODEL dtype).  ATen CPU computes every fp16/bf16 elementwise op in fp32 and rounds the result to
   the tensor dtype; reductions accumulate in fp32 and round once.  rmode: 0 fp32, 1 fp16, 2 bf16. */
static inline float rnd(float v, int rmode) {
    if (rmode == 1) return gqo_f16_to_f32(gqo_f32_to_f16(v));
    if (rmode == 2) return gqo_bf16_to_f32(gqo_f32_to_bf16(v));
    return v;
}
#define R(v) rnd((v), rmode)

static inline float clampf(float v, float lo, float hi) {
    /* torch.clamp(min,max) = min(max(v, lo), hi); NaN propagates */
    if (v != v) return v;
    v = v < lo ? lo : v;
    v = v > hi ? hi : v;
    return v;
}

/* -------------------------------------------------------- make_k_quants */

#define GQO_MAXG 32

typedef struct {
    float w[GQO_MAXG]; /* weights = av_x + |x|           quant_utils.py:205 */
    float x_min, x_max;
    float sum_w, sum_x;
    float best_scale, best_err;
    int is_const;
    /* per-iteration scratch */
    float L[GQO_MAXG];
    float this_scale, this_min, cand_err;
    int valid;
} kq_state_t;

/* Checker-side instrumentation (not part of the reference): how often the panel-wide `continue` of :251-252 skipped an
   iteration whose candidate SOME group would have taken -- the event that makes a row slice of a matrix decide differently
   from the whole matrix (the HIP library reports the same count through gq_gptq_quantize_slice). */
static int64_t g_panel_researches = 0;
int64_t gqo_panel_researches(int reset) {
    int64_t v = g_panel_researches;
    if (reset) g_panel_researches = 0;
    return v;
}

static void make_k_quants_r(const float* x, int64_t n_groups, int G, int bits, double rmin, double rdelta, int nstep,
                            int rmode, float* scale_out, float* zero_out) {
    const float maxq = (float)((1 << bits) - 1);
    const float eps = R(1e-9f); /* quant_utils.py:69; rounds to 0 in fp16 */
    kq_state_t* st = (kq_state_t*)malloc(sizeof(kq_state_t) * (size_t)n_groups);
    float tmp[GQO_MAXG];

#pragma omp parallel for private(tmp) schedule(static)
    for (int64_t g = 0; g < n_groups; ++g) {
        const float* xg = x + g * G;
        kq_state_t* s = &st[g];
        /* :203-205 */
        for (int j = 0; j < G; ++j) tmp[j] = R(xg[j] * xg[j]);
        float sum_x2 = R(gqo_aten_sum(tmp, G));
        float av_x = R(sqrtf(R(sum_x2 / (float)G)));
        for (int j = 0; j < G; ++j) s->w[j] = R(av_x + fabsf(xg[j]));
        /* :208-211 */
        float mn = xg[0], mx = xg[0];
        for (int j = 1; j < G; ++j) {
            mn = xg[j] < mn ? xg[j] : mn;
            mx = xg[j] > mx ? xg[j] : mx;
        }
        mn = mn < 0.0f ? mn : 0.0f; /* torch.minimum(x_min, 0): returns +0.0 on tie */
        s->x_min = mn;
        s->x_max = mx;
        s->is_const = (mx == mn);
        /* :214-215 */
        s->sum_w = R(gqo_aten_sum(s->w, G));
        for (int j = 0; j < G; ++j) tmp[j] = R(s->w[j] * xg[j]);
        s->sum_x = R(gqo_aten_sum(tmp, G));
        /* :218-220 */
        float sc = R(R(mx - mn) / maxq);
        if (s->is_const) sc = 0.0f;
        float isc = R(1.0f / (sc < eps ? eps : sc));
        /* :223-225, :228-232 */
        for (int j = 0; j < G; ++j) {
            float q = 0.0f;
            if (!s->is_const) {
                q = clampf(rintf(R(R(xg[j] - mn) * isc)), 0.0f, maxq);
                q = (float)(uint8_t)q; /* .to(torch.uint8) */
            }
            float diff = R(R(R(sc * q) + mn) - xg[j]);
            tmp[j] = R(s->w[j] * R(diff * diff));
        }
        s->best_scale = sc;
        s->best_err = R(gqo_aten_sum(tmp, G));
    }

    if (nstep >= 1) { /* :235-237 */
        for (int i = 0; i <= nstep; ++i) { /* :240 */
            /* python double arithmetic, rounded once to fp32 when multiplied (opmath scalar) */
            const float num = (float)(rmin + rdelta * (double)i + (double)maxq);
            int any_valid = 0;
#pragma omp parallel for private(tmp) reduction(| : any_valid) schedule(static)
            for (int64_t g = 0; g < n_groups; ++g) {
                const float* xg = x + g * G;
                kq_state_t* s = &st[g];
                /* :241  scalar / tensor == tensor.reciprocal() * scalar.
                   x_min here is the ALIASED best_min (:228, :270). */
                float den = R(s->x_max - s->x_min);
                den = den < eps ? eps : den;
                float cand_iscale = R(R(1.0f / den) * num);
                float t_l[GQO_MAXG], t_l2[GQO_MAXG], t_xl[GQO_MAXG];
                for (int j = 0; j < G; ++j) {
                    uint8_t qi = 0;
                    if (!s->is_const) { /* :243 */
                        float q = clampf(rintf(R(R(xg[j] - s->x_min) * cand_iscale)), 0.0f, maxq);
                        qi = (uint8_t)q; /* :242 */
                    }
                    uint8_t q2 = (uint8_t)(qi * qi); /* :246 new_q**2 stays uint8 (wraps) */
                    float qf = (float)qi;
                    s->L[j] = qf;
                    t_l[j] = R(s->w[j] * qf);                 /* :245 */
                    t_l2[j] = R(s->w[j] * (float)q2);         /* :246 */
                    t_xl[j] = R(R(s->w[j] * xg[j]) * qf);     /* :247 */
                }
                float sum_l = R(gqo_aten_sum(t_l, G));
                float sum_l2 = R(gqo_aten_sum(t_l2, G));
                float sum_xl = R(gqo_aten_sum(t_xl, G));
                float D = R(R(s->sum_w * sum_l2) - R(sum_l * sum_l)); /* :249 */
                s->valid = D > eps;                                    /* :250 */
                any_valid |= s->valid;
                float this_scale = R(R(R(s->sum_w * sum_xl) - R(s->sum_x * sum_l)) / D); /* :254 */
                float this_min = R(R(R(sum_l2 * s->sum_x) - R(sum_l * sum_xl)) / D);     /* :255 */
                if (this_min > 0.0f) { /* :257-260 */
                    float c = sum_l2 < eps ? eps : sum_l2;
                    this_scale = R(sum_xl / c);
                    this_min = 0.0f;
                }
                for (int j = 0; j < G; ++j) { /* :262-264 */
                    float diff = R(R(R(this_scale * s->L[j]) + this_min) - xg[j]);
                    tmp[j] = R(s->w[j] * R(diff * diff));
                }
                s->cand_err = R(gqo_aten_sum(tmp, G));
                s->this_scale = this_scale;
                s->this_min = this_min;
            }
            if (!any_valid) { /* :251-252: panel-wide early continue */
                for (int64_t g = 0; g < n_groups; ++g)
                    if (st[g].cand_err < st[g].best_err) {
                        g_panel_researches += 1;
                        break;
                    }
                continue;
            }
#pragma omp parallel for schedule(static)
            for (int64_t g = 0; g < n_groups; ++g) {
                kq_state_t* s = &st[g];
                if (s->cand_err < s->best_err) { /* :266-271 (NaN compares false) */
                    s->best_err = s->cand_err;
                    s->best_scale = s->this_scale;
                    s->x_min = s->this_min; /* best_min IS x_min (:228 alias) */
                }
            }
        }
    }
    for (int64_t g = 0; g < n_groups; ++g) { /* :273-274 */
        scale_out[g] = st[g].best_scale;
        zero_out[g] = -st[g].x_min;
    }
    free(st);
}

void gqo_make_k_quants(const float* x, int64_t n_groups, int G, int bits,
                       double rmin, double rdelta, int nstep, float* scale_out, float* zero_out) {
    make_k_quants_r(x, n_groups, G, bits, rmin, rdelta, nstep, 0, scale_out, zero_out);
}

/* ---------------------------------------------------------- make_quants */

/* quant_scale == "mse" (quant_utils.py:164-191): set through gqo_set_quant_scale (test infrastructure: one global). */
static int g_mse = 0, g_grid = 100;
static double g_maxshrink = 0.8;
void gqo_set_quant_scale(int mse, int grid, double maxshrink) {
    g_mse = mse;
    g_grid = grid;
    g_maxshrink = maxshrink;
}

static void make_quants_r(const float* x, int64_t n_groups, int G, int bits, int rmode, float* scale, float* zero) {
    const float maxq = (float)((1 << bits) - 1); /* quant_utils.py:74 */
#pragma omp parallel for schedule(static)
    for (int64_t g = 0; g < n_groups; ++g) {
        const float* xg = x + g * G;
        float mn = xg[0], mx = xg[0];
        for (int j = 1; j < G; ++j) {
            mn = xg[j] < mn ? xg[j] : mn;
            mx = xg[j] > mx ? xg[j] : mx;
        }
        float a = fabsf(mn);
        mx = a > mx ? a : mx;      /* :153 torch.maximum(|xmin|, xmax) */
        if (mn < 0.0f) mn = -mx;   /* :154-156 */
        if (mn == mx) {            /* :157-159 */
            mn = -1.0f;
            mx = 1.0f;
        }
        scale[g] = R(R(mx - mn) / maxq); /* :161 */
        zero[g] = 0.0f;                  /* :195 */
        if (g_mse) { /* :164-191, verbatim incl. the .round() that lands on the SCALE (:180) and the un-rounded q_int */
            const float zq = R((maxq + 1.0f) / 2.0f); /* :162 */
            const double den = g_maxshrink * (double)g_grid;
            const int n = (int)den + 1; /* :169 */
            float min_loss = INFINITY, best = 0.0f;
            float amax = fabsf(mn);
            amax = mx > amax ? mx : amax; /* :171 torch.max(xmax, |xmin|) */
            for (int i = 0; i < n; ++i) {
                const float alpha = R((float)(1.0 - (double)i / den)); /* :170, python double -> tensor dtype */
                const float cand = R(amax * alpha);
                const float xmax1 = mx < cand ? mx : cand;     /* :173 */
                const float xmin1 = mn > -cand ? mn : -cand;   /* :174 */
                const float scale1 = R(R(xmax1 - xmin1) / maxq); /* :176 */
                float c = scale1 < R(1e-9f) ? R(1e-9f) : scale1;
                if (scale1 != scale1) c = scale1;
                const float dv = rintf(c); /* :180 clamp_min(1e-9).round() */
                float terms[32];
                for (int j = 0; j < G; ++j) {
                    float q = clampf(R(R(xg[j] - zq) / dv), 0.0f, maxq); /* :179-181 */
                    float y = R(R(q * scale1) + zq);                      /* :182 */
                    float df = R(y - xg[j]);
                    terms[j] = R(df * df);                                /* :183 pow(2.0) == x * x */
                }
                const float loss = R(gqo_aten_sum(terms, G));
                if (loss < min_loss) { /* :185-189 */
                    min_loss = loss;
                    best = scale1;
                }
            }
            scale[g] = best; /* :190 */
        }
    }
}

void gqo_make_quants(const float* x, int64_t n_groups, int G, int bits, float* scale, float* zero) {
    make_quants_r(x, n_groups, G, bits, 0, scale, zero);
}

/* --------------------------------------------------------- scale search */

static void finish_super(const float* gscale, const float* gzero, int ng, int scale_maxq,
                         uint16_t* d, uint8_t* s, uint16_t* dmin, uint8_t* m, int is_signed, int rmode) {
    /* quant_utils.py:121-143 for one row */
    float max_scale = gscale[0], max_zero = gzero[0];
    for (int j = 1; j < ng; ++j) {
        max_scale = gscale[j] > max_scale ? gscale[j] : max_scale;
        max_zero = gzero[j] > max_zero ? gzero[j] : max_zero;
    }
    const float smq = (float)scale_maxq;
    *d = gqo_f32_to_f16(R(max_scale / smq));   /* :124 true divide, then .to(float16) */
    *dmin = gqo_f32_to_f16(R(max_zero / smq)); /* :125 */
    /* :128-129  int / tensor == tensor.reciprocal() * int */
    float inv_scale = max_scale > 0.0f ? R(R(1.0f / max_scale) * smq) : 0.0f;
    float inv_zero = max_zero > 0.0f ? R(R(1.0f / max_zero) * smq) : 0.0f;
    for (int j = 0; j < ng; ++j) { /* :132-143 */
        float a = clampf(rintf(R(inv_scale * gscale[j])), 0.0f, smq);
        float b = clampf(rintf(R(inv_zero * gzero[j])), 0.0f, smq);
        if (is_signed) {
            s[j] = (uint8_t)(int8_t)a;
            m[j] = (uint8_t)(int8_t)b;
        } else {
            s[j] = (uint8_t)a;
            m[j] = (uint8_t)b;
        }
    }
}

static void scale_search_r(const float* x, int64_t rows, int64_t ld, int q_type,
                           double rmin, double rdelta, int nstep, int rmode,
                           uint16_t* d, int64_t d_stride, uint8_t* s, int64_t s_ld,
                           uint16_t* dmin, int64_t dmin_stride, uint8_t* m, int64_t m_ld) {
    gqo_type_info_t ti;
    if (gqo_type_info(q_type, &ti)) return;
    const int G = ti.group, ng = 256 / G;
    /* :112-114 contiguous [rows*ng, G] copy of the strided panel */
    float* xc = (float*)malloc(sizeof(float) * (size_t)rows * 256);
    float* gs = (float*)malloc(sizeof(float) * (size_t)rows * ng);
    float* gz = (float*)malloc(sizeof(float) * (size_t)rows * ng);
    for (int64_t r = 0; r < rows; ++r) memcpy(xc + r * 256, x + r * ld, 256 * sizeof(float));
    if (ti.k_search)
        make_k_quants_r(xc, rows * ng, G, ti.bits, rmin, rdelta, nstep, rmode, gs, gz);
    else
        make_quants_r(xc, rows * ng, G, ti.bits, rmode, gs, gz);
    for (int64_t r = 0; r < rows; ++r)
        finish_super(gs + r * ng, gz + r * ng, ng, ti.scale_maxq, d + r * d_stride, s + r * s_ld,
                     dmin + r * dmin_stride, m + r * m_ld, ti.is_signed, rmode);
    free(xc);
    free(gs);
    free(gz);
}

void gqo_scale_search(const float* x, int64_t rows, int64_t ld, int q_type,
                      double rmin, double rdelta, int nstep,
                      uint16_t* d, int64_t d_stride, uint8_t* s, int64_t s_ld,
                      uint16_t* dmin, int64_t dmin_stride, uint8_t* m, int64_t m_ld) {
    scale_search_r(x, rows, ld, q_type, rmin, rdelta, nstep, 0, d, d_stride, s, s_ld, dmin, dmin_stride, m, m_ld);
}

/* --------------------------------------------------- quantize/dequantize */

static inline float ival(uint8_t b, int is_signed) { return is_signed ? (float)(int8_t)b : (float)b; }

float gqo_quantize1(float x, uint16_t d, int s, uint16_t dmin, int m, int qmin, int qmax) {
    /* quant_utils.py:34-40 */
    float ds = gqo_f16_to_f32(d) * (float)s;
    float dm = gqo_f16_to_f32(dmin) * (float)m;
    ds = ds < 1e-9f ? 1e-9f : ds; /* clamp_min(eps) (NaN-free inputs) */
    float q = rintf((x + dm) / ds);
    return clampf(q, (float)qmin, (float)qmax);
}

float gqo_dequantize1(float q, uint16_t d, int s, uint16_t dmin, int m) {
    /* quant_utils.py:43-46 */
    float ds = gqo_f16_to_f32(d) * (float)s;
    float dm = gqo_f16_to_f32(dmin) * (float)m;
    return ds * q - dm;
}

/* --------------------------------------------------- EvoPress FastOBQ step (uniform grids) */

/* evopress/src/quant_utils.py:57-106 Quantizer.find_params(x, weight=True) with perchannel=True on a [R, G] panel
   (row stride ld): per row xmin / xmax, sym: symmetric range; equal -> (-1, +1); scale = (xmax - xmin) / maxq;
   zero = (maxq + 1) / 2 (sym) or round(-xmin / scale). */
void gqo_uniform_params(const float* x, int64_t rows, int64_t ld, int G, int bits, int sym, float* scale, int64_t s_ld,
                        float* zero, int64_t z_ld) {
    const float maxq = (float)((1 << bits) - 1);
    for (int64_t r = 0; r < rows; ++r) {
        const float* xr = x + r * ld;
        float mn = xr[0], mx = xr[0];
        for (int j = 1; j < G; ++j) {
            mn = xr[j] < mn ? xr[j] : mn;
            mx = xr[j] > mx ? xr[j] : mx;
        }
        if (sym) {
            float a = fabsf(mn);
            mx = a > mx ? a : mx;
            if (mn < 0.0f) mn = -mx;
        }
        if (mn == mx) {
            mn = -1.0f;
            mx = 1.0f;
        }
        const float sc = (mx - mn) / maxq;
        scale[r * s_ld] = sc;
        zero[r * z_ld] = sym ? (maxq + 1.0f) / 2.0f : rintf(-mn / sc);
    }
}

/* evopress/src/fast_obq.py:131-200 for ONE bit width, given U = chol_upper(H^-1): the GPTQ column loop with
   q = clamp(round(w / max(scale, 1e-9) + zero), 0, maxq), w_hat = scale * (q - zero) (quant_utils.py:23-29) and the
   grid of a group found lazily from the current w at the group's first column (:168-171).  group_size == 0: one
   grid per row from the ORIGINAL w (:153-154); the reference leaves its scale / zero OUTPUTS uninitialised then
   (:157-159 only keep them on the handle) -- here they are returned in column 0.  W becomes the dequantized matrix. */
void gqo_obq_step(float* W, const float* U, int64_t R, int64_t C, int bits, int group_size, int sym, int block_size,
                  uint8_t* qweight, float* scale, float* zero) {
    const float maxq = (float)((1 << bits) - 1);
    const int64_t G = group_size > 0 ? group_size : C, ng = C / G;
    if (block_size <= 0) block_size = (int)C;
    if (group_size <= 0) gqo_uniform_params(W, R, C, (int)C, bits, sym, scale, ng, zero, ng);
    float* w_blk = (float*)malloc(sizeof(float) * (size_t)R * block_size);
    float* errs = (float*)malloc(sizeof(float) * (size_t)R * block_size);
    for (int64_t c1 = 0; c1 < C; c1 += block_size) {
        int64_t c2 = c1 + block_size < C ? c1 + block_size : C;
        int ncols = (int)(c2 - c1);
        for (int64_t r = 0; r < R; ++r) memcpy(w_blk + r * ncols, W + r * C + c1, sizeof(float) * ncols);
        for (int i = 0; i < ncols; ++i) {
            const int64_t col = c1 + i, g = col / G;
            if (group_size > 0 && col % G == 0) /* :168-171 reads w, not w_blk */
                gqo_uniform_params(W + col, R, C, (int)G, bits, sym, scale + g, ng, zero + g, ng);
            const float dii = U[col * C + col];
            const float* urow = U + col * C + c1;
#pragma omp parallel for schedule(static)
            for (int64_t r = 0; r < R; ++r) {
                float* wb = w_blk + r * ncols;
                const float sc = scale[r * ng + g], zp = zero[r * ng + g];
                const float wci = wb[i];
                const float q = clampf(rintf(wci / (sc < 1e-9f ? 1e-9f : sc) + zp), 0.0f, maxq);
                const float wq = sc * (q - zp);
                qweight[r * C + col] = (uint8_t)q;
                const float err = (wci - wq) / dii;
                W[r * C + col] = wq;
                const float nerr = -1.0f * err;
                for (int j = i; j < ncols; ++j) wb[j] = wb[j] + nerr * urow[j];
                errs[r * ncols + i] = err;
            }
        }
        if (c2 < C) { /* :197 addmm_(errs, H_inv_cho[c1:c2, c2:], alpha=-1): k-ordered fma chain, one subtraction */
#pragma omp parallel
            {
                float* acc = (float*)malloc(sizeof(float) * (size_t)(C - c2));
#pragma omp for schedule(static)
                for (int64_t r = 0; r < R; ++r) {
                    const float* e = errs + r * ncols;
                    float* wr = W + r * C + c2;
                    const int64_t n = C - c2;
                    for (int64_t j = 0; j < n; ++j) acc[j] = 0.0f;
                    for (int k = 0; k < ncols; ++k) {
                        const float ek = e[k];
                        const float* ur = U + (c1 + k) * C + c2;
                        for (int64_t j = 0; j < n; ++j) acc[j] = fmaf(ek, ur[j], acc[j]);
                    }
                    for (int64_t j = 0; j < n; ++j) wr[j] = wr[j] - acc[j];
                }
                free(acc);
            }
        }
    }
    free(w_blk);
    free(errs);
}

/* ------------------------------------------------------------ GPTQ step */

/* perm != NULL: act_order (gptq.py:208-216): W/U are in permuted order, d/s/dmin/m hold the static scales of
   the ORIGINAL column groups (inputs), column `col` uses the groups of original column perm[col]
   (group_idx / super_group_idx of gptq.py:215-216, looked up at :233-235). */
static void gptq_step_impl(float* W, const float* U, int64_t R, int64_t C, int q_type,
                           int block_size, int static_groups,
                           double rmin, double rdelta, int nstep,
                           uint8_t* qweight, uint16_t* d, uint8_t* s, uint16_t* dmin, uint8_t* m,
                           const int32_t* perm) {
    gqo_type_info_t ti;
    if (gqo_type_info(q_type, &ti)) return;
    const int G = ti.group;
    const int64_t ng = C / G, nsg = C / 256;
    const int gps = 256 / G;
    if (block_size <= 0) block_size = (int)C; /* gptq.py:54 */
    if (q_type == GQO_Q3_K) static_groups = 0; /* gptq.py:204-206 */
    if (perm) static_groups = 2;

    if (static_groups == 1) { /* gptq.py:184-196 */
        for (int64_t c = 0; c < C; c += 256)
            gqo_scale_search(W + c, R, C, q_type, rmin, rdelta, nstep, d + c / 256, nsg,
                             s + (c / 256) * gps, ng, dmin + c / 256, nsg, m + (c / 256) * gps, ng);
    }

    float* w_blk = (float*)malloc(sizeof(float) * (size_t)R * block_size);
    float* errs = (float*)malloc(sizeof(float) * (size_t)R * block_size);

    for (int64_t c1 = 0; c1 < C; c1 += block_size) { /* gptq.py:222 */
        int64_t c2 = c1 + block_size < C ? c1 + block_size : C;
        int ncols = (int)(c2 - c1);
        for (int64_t r = 0; r < R; ++r)
            memcpy(w_blk + r * ncols, W + r * C + c1, sizeof(float) * ncols); /* :225 clone */
        for (int i = 0; i < ncols; ++i) { /* :229 */
            int64_t col = c1 + i;
            int64_t g_idx = (perm ? perm[col] : col) / G, sg_idx = (perm ? perm[col] : col) / 256; /* :233-238 */
            if (!static_groups && (col % 256) == 0) /* :240-245 reads w, NOT w_blk */
                gqo_scale_search(W + col, R, C, q_type, rmin, rdelta, nstep, d + sg_idx, nsg,
                                 s + g_idx, ng, dmin + sg_idx, nsg, m + g_idx, ng);
            const float dii = U[col * C + col];
            const float* urow = U + col * C + c1; /* H_inv_cho_blk[i, :] */
#pragma omp parallel for schedule(static)
            for (int64_t r = 0; r < R; ++r) {
                float* wb = w_blk + r * ncols;
                uint16_t dd = d[r * nsg + sg_idx], dm = dmin[r * nsg + sg_idx];
                int si = (int)ival(s[r * ng + g_idx], ti.is_signed);
                int mi = (int)ival(m[r * ng + g_idx], ti.is_signed);
                float wci = wb[i];
                float q = gqo_quantize1(wci, dd, si, dm, mi, ti.qmin, ti.qmax); /* :247-254 */
                float wq = gqo_dequantize1(q, dd, si, dm, mi);                    /* :255-261 */
                qweight[r * C + col] = ti.is_signed ? (uint8_t)(int8_t)q : (uint8_t)q; /* :263 */
                float err = (wci - wq) / dii; /* :264 */
                W[r * C + col] = wq;          /* :266 */
                /* :267 addr_(err, U[i, i:], alpha=-1): ATen CPU evaluates
                   self + (alpha*err)*u with the mul and the add as separate
                   roundings (see test_oracle_golden G6). */
                float nerr = -1.0f * err;
                for (int j = i; j < ncols; ++j) wb[j] = wb[j] + nerr * urow[j];
                errs[r * ncols + i] = err; /* :268 */
            }
        }
        /* :270  w[:, c2:] -= errs @ U[c1:c2, c2:].  Restated as: per output
           element a k-ordered fp32 fma chain from 0 over the block's ncols
           columns, then one subtraction.  Bit-equal to MKL sgemm on the build
           container for K<=128 is NOT guaranteed; golden G6 reports the rate. */
        if (c2 < C) {
            /* k outermost per row: U is read along its rows (the j-inner loop strides by a whole row of U, 16 KiB at
               C = 4096 -- every load of a column hits the same cache set and, when W and U happen to share their
               address bits below 4 KiB, aliases the pending stores: 35 s instead of 1.6 s per 1024 x 4096 on the GPU
               box's host).  Per element still acc = fmaf(e[k], U[c1+k][j], acc) for k = 0.., then one subtraction. */
#pragma omp parallel
            {
                float* acc = (float*)malloc(sizeof(float) * (size_t)(C - c2));
#pragma omp for schedule(static)
                for (int64_t r = 0; r < R; ++r) {
                    const float* e = errs + r * ncols;
                    float* wr = W + r * C + c2;
                    const int64_t n = C - c2;
                    for (int64_t j = 0; j < n; ++j) acc[j] = 0.0f;
                    for (int k = 0; k < ncols; ++k) {
                        const float ek = e[k];
                        const float* urow = U + (c1 + k) * C + c2;
                        for (int64_t j = 0; j < n; ++j) acc[j] = fmaf(ek, urow[j], acc[j]);
                    }
                    for (int64_t j = 0; j < n; ++j) wr[j] = wr[j] - acc[j];
                }
                free(acc);
            }
        }
    }
    free(w_blk);
    free(errs);
}

void gqo_gptq_step(float* W, const float* U, int64_t R, int64_t C, int q_type,
                   int block_size, int static_groups,
                   double rmin, double rdelta, int nstep,
                   uint8_t* qweight, uint16_t* d, uint8_t* s, uint16_t* dmin, uint8_t* m) {
    gptq_step_impl(W, U, R, C, q_type, block_size, static_groups, rmin, rdelta, nstep, qweight, d, s, dmin, m, NULL);
}

void gqo_gptq_step_perm(float* W, const float* U, int64_t R, int64_t C, int q_type, int block_size,
                        const int32_t* perm, const uint16_t* d, const uint8_t* s, const uint16_t* dmin,
                        const uint8_t* m, uint8_t* qweight) {
    gptq_step_impl(W, U, R, C, q_type, block_size, 1, 0.0, 0.0, 0, qweight, (uint16_t*)d, (uint8_t*)s,
                   (uint16_t*)dmin, (uint8_t*)m, perm);
}

/* ------------------------------------------------------------------ RTN */

void gqo_rtn_quantize(const float* W, int64_t R, int64_t C, int q_type,
                      double rmin, double rdelta, int nstep,
                      uint8_t* qweight, uint16_t* d, uint8_t* s, uint16_t* dmin, uint8_t* m) {
    /* quantizer.py:278-330 with fp32 weights */
    gqo_type_info_t ti;
    if (gqo_type_info(q_type, &ti)) return;
    const int G = ti.group, gps = 256 / G;
    const int64_t ng = C / G, nsg = C / 256;
    for (int64_t c = 0; c < C; c += 256)
        gqo_scale_search(W + c, R, C, q_type, rmin, rdelta, nstep, d + c / 256, nsg,
                         s + (c / 256) * gps, ng, dmin + c / 256, nsg, m + (c / 256) * gps, ng);
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < R; ++r)
        for (int64_t c = 0; c < C; ++c) {
            float q = gqo_quantize1(W[r * C + c], d[r * nsg + c / 256],
                                    (int)ival(s[r * ng + c / G], ti.is_signed), dmin[r * nsg + c / 256],
                                    (int)ival(m[r * ng + c / G], ti.is_signed), ti.qmin, ti.qmax);
            qweight[r * C + c] = ti.is_signed ? (uint8_t)(int8_t)q : (uint8_t)q;
        }
}

void gqo_rtn_quantize_lp(const float* W, int rmode, int64_t R_, int64_t C, int q_type,
                         double rmin, double rdelta, int nstep,
                         uint8_t* qweight, uint16_t* d, uint8_t* s, uint16_t* dmin, uint8_t* m) {
    /* quantizer.py:278-330 with fp16 (rmode 1) / bf16 (rmode 2) weights: W holds the weight VALUES widened
       to fp32; make_*quants and the super-group step run in the model dtype, the final quantize() in fp32. */
    gqo_type_info_t ti;
    if (gqo_type_info(q_type, &ti)) return;
    const int G = ti.group, gps = 256 / G;
    const int64_t ng = C / G, nsg = C / 256;
    for (int64_t c = 0; c < C; c += 256)
        scale_search_r(W + c, R_, C, q_type, rmin, rdelta, nstep, rmode, d + c / 256, nsg, s + (c / 256) * gps, ng,
                       dmin + c / 256, nsg, m + (c / 256) * gps, ng);
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < R_; ++r)
        for (int64_t c = 0; c < C; ++c) {
            float q = gqo_quantize1(W[r * C + c], d[r * nsg + c / 256], (int)ival(s[r * ng + c / G], ti.is_signed),
                                    dmin[r * nsg + c / 256], (int)ival(m[r * ng + c / G], ti.is_signed), ti.qmin,
                                    ti.qmax);
            qweight[r * C + c] = ti.is_signed ? (uint8_t)(int8_t)q : (uint8_t)q;
        }
}

/* ----------------------------------------------------------- dequantize */

void gqo_dequantize(int q_type, const uint8_t* qweight, const uint16_t* d, const uint8_t* s,
                    const uint16_t* dmin, const uint8_t* m, int64_t R, int64_t C, float* out) {
    gqo_type_info_t ti;
    if (gqo_type_info(q_type, &ti)) return;
    const int G = ti.group;
    const int64_t ng = C / G, nsg = C / 256;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < R; ++r)
        for (int64_t c = 0; c < C; ++c)
            out[r * C + c] = gqo_dequantize1(ival(qweight[r * C + c], ti.is_signed), d[r * nsg + c / 256],
                                             (int)ival(s[r * ng + c / G], ti.is_signed),
                                             dmin[r * nsg + c / 256],
                                             (int)ival(m[r * ng + c / G], ti.is_signed));
}

/* -------------------------------------------------------------- packers */

static void pack_scale_min(const uint8_t* sc, const uint8_t* mn, uint8_t* out) {
    /* packing_utils.py:8-30 */
    for (int j = 0; j < 4; ++j) {
        out[j] = (uint8_t)(sc[j] | ((sc[4 + j] >> 4) << 6));
        out[4 + j] = (uint8_t)(mn[j] | ((mn[4 + j] >> 4) << 6));
        out[8 + j] = (uint8_t)((sc[4 + j] & 0x0F) | ((mn[4 + j] & 0x0F) << 4));
    }
}

static inline void put16(uint8_t* p, uint16_t v) {
    p[0] = (uint8_t)(v & 0xff);
    p[1] = (uint8_t)(v >> 8);
}

int gqo_pack(int q_type, const uint8_t* qw, const uint16_t* d, const uint8_t* s,
             const uint16_t* dmin, const uint8_t* m, int64_t R, int64_t C, uint8_t* out) {
    gqo_type_info_t ti;
    if (gqo_type_info(q_type, &ti)) return -1;
    if (C % 256) return -2;
    const int64_t nb = R * (C / 256);
    const int gps = 256 / ti.group;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < nb; ++b) {
        const uint8_t* q = qw + b * 256;
        const uint8_t* sb = s + b * gps;
        const uint8_t* mb = m ? m + b * gps : NULL;
        uint8_t* o = out + b * ti.type_size;
        uint8_t v[256];
        switch (q_type) {
        case GQO_Q2_K: { /* packing_utils.py:33-77 : scales[16] qs[64] d dmin */
            for (int j = 0; j < 16; ++j) o[j] = (uint8_t)((sb[j] & 0x0F) | ((mb[j] & 0x0F) << 4));
            for (int ch = 0; ch < 2; ++ch)
                for (int l = 0; l < 32; ++l) {
                    const uint8_t* c = q + ch * 128;
                    o[16 + ch * 32 + l] =
                        (uint8_t)(c[l] | (c[32 + l] << 2) | (c[64 + l] << 4) | (c[96 + l] << 6));
                }
            put16(o + 80, d[b]);
            put16(o + 82, dmin[b]);
        } break;
        case GQO_Q3_K: { /* :80-142 : hmask[32] qs[64] scales[12] d */
            uint8_t sc[16];
            for (int j = 0; j < 256; ++j) v[j] = (uint8_t)((int8_t)q[j] + 4);  /* :94 */
            for (int j = 0; j < 16; ++j) sc[j] = (uint8_t)((int8_t)sb[j] + 32); /* :95 */
            memset(o, 0, 110);
            for (int j = 0; j < 256; ++j) { /* :119-125 */
                if (v[j] > 3) {
                    o[j % 32] |= (uint8_t)(1 << (j / 32));
                    v[j] = (uint8_t)(v[j] - 4);
                }
            }
            for (int ch = 0; ch < 2; ++ch)
                for (int l = 0; l < 32; ++l) {
                    const uint8_t* c = v + ch * 128;
                    o[32 + ch * 32 + l] =
                        (uint8_t)(c[l] | (c[32 + l] << 2) | (c[64 + l] << 4) | (c[96 + l] << 6));
                }
            uint8_t* sbytes = o + 96;
            for (int j = 0; j < 16; ++j) { /* :103-115 */
                uint8_t lo4 = sc[j] & 0x0F, hi2 = (sc[j] >> 4) & 0x03;
                if (j < 8) sbytes[j] |= lo4;
                else sbytes[j - 8] |= (uint8_t)(lo4 << 4);
                sbytes[8 + (j % 4)] |= (uint8_t)(hi2 << (2 * (j / 4)));
            }
            put16(o + 108, d[b]);
        } break;
        case GQO_Q4_K: { /* :145-190 : d dmin scales[12] qs[128] */
            put16(o, d[b]);
            put16(o + 2, dmin[b]);
            pack_scale_min(sb, mb, o + 4);
            for (int base = 0; base < 256; base += 64)
                for (int l = 0; l < 32; ++l)
                    o[16 + (base / 64) * 32 + l] = (uint8_t)(q[base + l] | (q[base + 32 + l] << 4));
        } break;
        case GQO_Q5_K: { /* :193-262 : d dmin scales[12] qh[32] ql[128] */
            put16(o, d[b]);
            put16(o + 2, dmin[b]);
            pack_scale_min(sb, mb, o + 4);
            memset(o + 16, 0, 32);
            for (int base = 0, k = 0; base < 256; base += 64, ++k)
                for (int j = 0; j < 32; ++j) {
                    uint8_t l1 = q[base + j], l2 = q[base + j + 32];
                    if (l1 > 15) { o[16 + j] |= (uint8_t)(1 << (2 * k)); l1 = (uint8_t)(l1 - 16); }
                    if (l2 > 15) { o[16 + j] |= (uint8_t)(2 << (2 * k)); l2 = (uint8_t)(l2 - 16); }
                    o[48 + j + base / 2] = (uint8_t)(l1 | (l2 << 4));
                }
        } break;
        case GQO_Q6_K: { /* :265-326 : ql[128] qh[64] scales[16] d */
            for (int j = 0; j < 256; ++j) v[j] = (uint8_t)((int8_t)q[j] + 32); /* :279 */
            for (int ch = 0; ch < 2; ++ch)
                for (int l = 0; l < 32; ++l) {
                    const uint8_t* c = v + ch * 128;
                    uint8_t v0 = c[l], v1 = c[l + 32], v2 = c[l + 64], v3 = c[l + 96];
                    o[ch * 64 + l] = (uint8_t)((v0 & 0xF) | ((v2 & 0xF) << 4));
                    o[ch * 64 + 32 + l] = (uint8_t)((v1 & 0xF) | ((v3 & 0xF) << 4));
                    o[128 + ch * 32 + l] = (uint8_t)(((v0 >> 4) & 3) | (((v1 >> 4) & 3) << 2) |
                                                     (((v2 >> 4) & 3) << 4) | (((v3 >> 4) & 3) << 6));
                }
            for (int j = 0; j < 16; ++j) o[192 + j] = sb[j]; /* :321 int8 -> uint8 view */
            put16(o + 208, d[b]);
        } break;
        }
    }
    return 0;
}

/* ------------------------------------------------------- Hessian pieces */

void gqo_h_accumulate(float* H, const float* X, int64_t T, int64_t C, float beta, float alpha) {
    /* gptq.py:108-112  H = beta*H + alpha * X^T X  (double accumulation here:
       the oracle is the accuracy anchor for a tolerance-class stage) */
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t i = 0; i < C; ++i)
        for (int64_t j = 0; j < C; ++j) {
            double acc = 0.0;
            for (int64_t t = 0; t < T; ++t) acc += (double)X[t * C + i] * (double)X[t * C + j];
            H[i * C + j] = (float)((double)beta * (double)H[i * C + j] + (double)alpha * acc);
        }
}

static void damp_diag(float* H, int64_t C, float rel_damp) {
    /* gptq.py:315-316 damping: mean of the diagonal in fp32 (ATen sums in its own order; tolerance-class) */
    double tr = 0.0;
    for (int64_t i = 0; i < C; ++i) tr += H[i * C + i];
    float damp = rel_damp * (float)(tr / (double)C);
    for (int64_t i = 0; i < C; ++i) H[i * C + i] += damp;
}

/* obq: EvoPress FastOBQ order (evopress/src/fast_obq.py:133-141 damp first, :221-228 then mask with an undamped 1) */
static int h_prepare_impl(float* H, float* W, int64_t R, int64_t C, float rel_damp, float* U, int obq) {
    /* gptq.py:134-135,141 */
    for (int64_t i = 0; i < C; ++i)
        if (H[i * C + i] == 0.0f) {
            H[i * C + i] = 1.0f;
            for (int64_t r = 0; r < R; ++r) W[r * C + i] = 0.0f;
        }
    if (obq) damp_diag(H, C, rel_damp);
    /* gptq.py:307-313 zero columns of W */
    for (int64_t j = 0; j < C; ++j) {
        int allz = 1;
        for (int64_t r = 0; r < R && allz; ++r) allz = (W[r * C + j] == 0.0f);
        if (allz) {
            for (int64_t k = 0; k < C; ++k) H[j * C + k] = 0.0f, H[k * C + j] = 0.0f;
            H[j * C + j] = 1.0f;
        }
    }
    if (!obq) damp_diag(H, C, rel_damp);

    /* :318-320  U = chol_upper(inv(H)) in double, via H = L L^T, Hinv = L^-T L^-1 */
    double* A = (double*)malloc(sizeof(double) * (size_t)C * C);
    double* Li = (double*)calloc((size_t)C * C, sizeof(double));
    int bad = 0;
    for (int64_t i = 0; i < C * C; ++i) A[i] = H[i];
    for (int64_t j = 0; j < C && !bad; ++j) { /* lower Cholesky in place */
        double sum = A[j * C + j];
        for (int64_t k = 0; k < j; ++k) sum -= A[j * C + k] * A[j * C + k];
        if (!(sum > 0.0)) { bad = 1; break; }
        double ljj = sqrt(sum);
        A[j * C + j] = ljj;
        for (int64_t i = j + 1; i < C; ++i) {
            double v = A[i * C + j];
            for (int64_t k = 0; k < j; ++k) v -= A[i * C + k] * A[j * C + k];
            A[i * C + j] = v / ljj;
        }
    }
    if (!bad) {
        /* Li = L^-1 (lower) */
        for (int64_t c = 0; c < C; ++c) {
            Li[c * C + c] = 1.0 / A[c * C + c];
            for (int64_t i = c + 1; i < C; ++i) {
                double v = 0.0;
                for (int64_t k = c; k < i; ++k) v -= A[i * C + k] * Li[k * C + c];
                Li[i * C + c] = v / A[i * C + i];
            }
        }
        /* Hinv = Li^T Li -> A (full symmetric) */
        for (int64_t i = 0; i < C; ++i)
            for (int64_t j = i; j < C; ++j) {
                double v = 0.0;
                for (int64_t k = j; k < C; ++k) v += Li[k * C + i] * Li[k * C + j];
                A[i * C + j] = v;
                A[j * C + i] = v;
            }
        /* upper Cholesky of Hinv: Hinv = U^T U; reuse Li as U storage */
        memset(Li, 0, sizeof(double) * (size_t)C * C);
        for (int64_t i = 0; i < C && !bad; ++i) {
            double sum = A[i * C + i];
            for (int64_t k = 0; k < i; ++k) sum -= Li[k * C + i] * Li[k * C + i];
            if (!(sum > 0.0)) { bad = 1; break; }
            double uii = sqrt(sum);
            Li[i * C + i] = uii;
            for (int64_t j = i + 1; j < C; ++j) {
                double v = A[i * C + j];
                for (int64_t k = 0; k < i; ++k) v -= Li[k * C + i] * Li[k * C + j];
                Li[i * C + j] = v / uii;
            }
        }
    }
    if (bad) { /* :321-323 identity fallback */
        for (int64_t i = 0; i < C * C; ++i) U[i] = 0.0f;
        for (int64_t i = 0; i < C; ++i) U[i * C + i] = 1.0f;
    } else {
        for (int64_t i = 0; i < C * C; ++i) U[i] = (float)Li[i];
    }
    free(A);
    free(Li);
    return bad;
}

int gqo_h_prepare(float* H, float* W, int64_t R, int64_t C, float rel_damp, float* U) {
    return h_prepare_impl(H, W, R, C, rel_damp, U, 0);
}
int gqo_obq_h_prepare(float* H, float* W, int64_t R, int64_t C, float rel_damp, float* U) {
    return h_prepare_impl(H, W, R, C, rel_damp, U, 1);
}

/* make_k_quants / make_quants in fp16 (rmode 1) / bf16 (rmode 2): exposed for tests */
void gqo_make_quants_lp(const float* x, int64_t n_groups, int G, int bits, int k_search, int rmode,
                        double rmin, double rdelta, int nstep, float* scale, float* zero) {
    if (k_search) make_k_quants_r(x, n_groups, G, bits, rmin, rdelta, nstep, rmode, scale, zero);
    else make_quants_r(x, n_groups, G, bits, rmode, scale, zero);
}
