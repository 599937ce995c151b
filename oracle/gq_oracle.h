/*
 * gq_oracle.h -- CPU ORACLE for the GPTQ -> GGUF K-quant hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  It is a plain-C restatement
 * of the reference algorithm (IST-DASLab/gptq-gguf-toolkit, quant/gptq/src/{gptq,quant_utils,packing_utils}.py)
 * and exists only so that tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg can check / time the HIP path against it.  Nothing under
 * gptq-gguf-toolkit_amd/ may include, link or call it.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit (or to the
 * stated tolerance for the fp32 linear-algebra stages) against outputs of the
 * reference itself, generated in the build container by
 * tests/golden/make_golden.py (imports /root/reference with a 3-constant stub
 * for the absent `gguf` package) and committed as tests/golden/ (npz files).
 *
 * All pointers are host pointers.  fp16 values travel as raw uint16_t bits.
 * Integer outputs (qweight, group scale/zero ints) are stored as int8_t-sized
 * bytes; unsigned types (Q2_K/Q4_K/Q5_K) hold 0..255 in the same byte.
 */
#ifndef GQ_ORACLE_H
#define GQ_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ggml type ids, reference quant_utils.py:11-16 */
enum { GQO_Q2_K = 10, GQO_Q3_K = 11, GQO_Q4_K = 12, GQO_Q5_K = 13, GQO_Q6_K = 14 };

typedef struct {
    int bits;       /* 2..6 */
    int qmin, qmax; /* clamp range of the stored ints */
    int scale_maxq; /* 15 / 31 / 63 */
    int group;      /* 16 or 32 */
    int is_signed;  /* 1: int8 outputs (Q3_K, Q6_K); 0: uint8 */
    int k_search;   /* 1: make_k_quants (Q2/Q4/Q5); 0: make_quants (Q3/Q6) */
    int type_size;  /* bytes per 256-value block in GGUF */
} gqo_type_info_t;

/* reference quant_utils.py:19-26 (+ gguf.constants.GGML_QUANT_SIZES) */
int gqo_type_info(int q_type, gqo_type_info_t* out);

/* fp16 helpers (IEEE binary16, round-to-nearest-even) */
uint16_t gqo_f32_to_f16(float f);
float gqo_f16_to_f32(uint16_t h);
uint16_t gqo_f32_to_bf16(float f);
float gqo_bf16_to_f32(uint16_t h);

/* ATen-CPU inner-dim fp32 sum order for n in {16,32} contiguous values
   (8 lanes, chunks added in order, lanes summed 0->7).  Exposed for tests. */
float gqo_aten_sum(const float* v, int n);

/* checker-side counter (see gq_oracle.c): skipped iterations (quant_utils.py:251-252) whose candidate a group would have taken */
int64_t gqo_panel_researches(int reset);

/* reference quant_utils.py:199-274 make_k_quants.  x: [n_groups, G] contiguous.
   scale/zero: [n_groups].  zero = -best_min. */
void gqo_make_k_quants(const float* x, int64_t n_groups, int G, int bits,
                       double rmin, double rdelta, int nstep, float* scale, float* zero);

/* reference quant_utils.py:147-197 make_quants, absmax branch. */
/* EvoPress FastOBQ (evopress/src/fast_obq.py:131-200, quant_utils.py:57-106): uniform grids */
void gqo_uniform_params(const float* x, int64_t rows, int64_t ld, int G, int bits, int sym, float* scale, int64_t s_ld,
                        float* zero, int64_t z_ld);
void gqo_obq_step(float* W, const float* U, int64_t R, int64_t C, int bits, int group_size, int sym, int block_size,
                  uint8_t* qweight, float* scale, float* zero);
/* quant_scale == "mse" for make_quants (quant_utils.py:164-191): 1 / grid / maxshrink; 0 = absmax (default) */
void gqo_set_quant_scale(int mse, int grid, double maxshrink);
void gqo_make_quants(const float* x, int64_t n_groups, int G, int bits, float* scale, float* zero);

/* reference quant_utils.py:90-145 get_scale_and_zero on a [rows,256] panel with
   row stride `ld` (elements).  Outputs: d[rows], dmin[rows] fp16 bits;
   s[rows, 256/G], m[rows, 256/G] ints (stride s_ld).  */
void gqo_scale_search(const float* x, int64_t rows, int64_t ld, int q_type,
                      double rmin, double rdelta, int nstep,
                      uint16_t* d, int64_t d_stride, uint8_t* s, int64_t s_ld,
                      uint16_t* dmin, int64_t dmin_stride, uint8_t* m, int64_t m_ld);

/* reference quant_utils.py:34-46, scalar forms */
float gqo_quantize1(float x, uint16_t d, int s, uint16_t dmin, int m, int qmin, int qmax);
float gqo_dequantize1(float q, uint16_t d, int s, uint16_t dmin, int m);

/* reference gptq.py:145-295 (rank-0 part of GPTQ.step), given the fp32 working
   copy W[R,C] (in/out: becomes the dequantized matrix) and U = chol_upper(H^-1).
   Outputs: qweight[R,C] bytes; d,dmin [R,C/256] fp16 bits; s,m [R,C/G] bytes.
   block_size as in the reference (0 => C).  static_groups as in gptq.py:184-196. */
void gqo_gptq_step(float* W, const float* U, int64_t R, int64_t C, int q_type,
                   int block_size, int static_groups,
                   double rmin, double rdelta, int nstep,
                   uint8_t* qweight, uint16_t* d, uint8_t* s, uint16_t* dmin, uint8_t* m);

/* GPTQ.step with act_order=True (gptq.py:208-216, 233-235): W, U already permuted by
   perm = argsort(diag(H), descending); d/s/dmin/m = static scales of the ORIGINAL column groups (inputs);
   qweight is produced in permuted positions (the reference un-permutes at :272-276). */
void gqo_gptq_step_perm(float* W, const float* U, int64_t R, int64_t C, int q_type, int block_size,
                        const int32_t* perm, const uint16_t* d, const uint8_t* s, const uint16_t* dmin,
                        const uint8_t* m, uint8_t* qweight);

/* reference quantizer.py:278-330 (_quant_non_block_module), fp32 weights. */
void gqo_rtn_quantize(const float* W, int64_t R, int64_t C, int q_type,
                      double rmin, double rdelta, int nstep,
                      uint8_t* qweight, uint16_t* d, uint8_t* s, uint16_t* dmin, uint8_t* m);

/* the same with the weight held in fp16 (rmode 1) or bf16 (rmode 2) -- the reference hands module.weight
   un-cast to get_scale_and_zero (quantizer.py:109,195), so make_*quants run in the model dtype.  W carries
   the weight values widened to fp32. */
void gqo_rtn_quantize_lp(const float* W, int rmode, int64_t R, int64_t C, int q_type,
                         double rmin, double rdelta, int nstep,
                         uint8_t* qweight, uint16_t* d, uint8_t* s, uint16_t* dmin, uint8_t* m);

/* reference quant_utils.py:277-310 dequantize_linear_weight -> fp32 [R,C] */
void gqo_dequantize(int q_type, const uint8_t* qweight, const uint16_t* d, const uint8_t* s,
                    const uint16_t* dmin, const uint8_t* m, int64_t R, int64_t C, float* out);

/* reference packing_utils.py:33-326.  Inputs are NOT mutated (the reference's
   pack_Q3K / pack_Q6K add +4/+32 in place; callers there pass clones).
   out: [R, C/256 * type_size] bytes. */
int gqo_pack(int q_type, const uint8_t* qweight, const uint16_t* d, const uint8_t* s,
             const uint16_t* dmin, const uint8_t* m, int64_t R, int64_t C, uint8_t* out);

/* reference gptq.py:79-114 GPTQ.update: H = beta*H + alpha * X^T X, X fp32 [T,C].
   Tolerance-class (summation order differs from MKL). */
void gqo_h_accumulate(float* H, const float* X, int64_t T, int64_t C, float beta, float alpha);

/* reference gptq.py:134-141 + 304-324: dead-channel fix, zero-column masking,
   damping, U = chol_upper(inv(H)).  Returns 0, or 1 if the identity fallback
   was taken (H not positive definite).  H is mutated like the reference. */
int gqo_h_prepare(float* H, float* W, int64_t R, int64_t C, float rel_damp, float* U);
/* EvoPress FastOBQ order: damping before the zero-column mask (evopress/src/fast_obq.py:133-141, 221-228) */
int gqo_obq_h_prepare(float* H, float* W, int64_t R, int64_t C, float rel_damp, float* U);

#ifdef __cplusplus
}
#endif
#endif
