/*
 * gptq_gguf.h -- C ABI of libgptqgguf_hip.so: the MI355X (gfx950) GPTQ -> GGUF
 * K-quant hot path.
 *
 * The reference (IST-DASLab/gptq-gguf-toolkit) has no FFI seam: its hot path is a
 * Python class protocol (quant/gptq/src/gptq.py GPTQ.update/quantize) over chains
 * of torch ops.  Each entry point below replaces one such chain; the citation
 * after "replaces" is the reference file:line (relative to quant/gptq/src/).
 * A maintainer of the reference would bind these with ctypes (see
 * INTEGRATION.md); gptq-gguf-toolkit_amd/_cabi.py is that binding.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (hipMalloc'd / torch.cuda storage) unless
 *    the parameter name ends in _host;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); every
 *    call only ENQUEUES work on that stream and returns; nothing is freed or
 *    allocated for the caller, scratch comes from the caller's `ws` buffer
 *    (size it with gq_workspace_bytes);
 *  - return value: 0 on success, negative gq_status on error (gq_last_error()
 *    returns a thread-local message);
 *  - fp16 fields (super-group scale d / min dmin) are raw IEEE binary16 bits;
 *  - integer outputs are one byte per value: uint8 for Q2_K/Q4_K/Q5_K, int8
 *    (two's complement in the same byte) for Q3_K/Q6_K, exactly the tensors the
 *    reference stores in data.pth (quantizer.py:268-275);
 *  - matrices are dense row-major: W[R,C] (nn.Linear weight, R = out features,
 *    C = in features, C % 256 == 0), H[C,C], U[C,C];
 *    qweight[R,C], d/dmin[R,C/256], s/m[R,C/G]  (G = 16 or 32).
 */
#ifndef GPTQ_GGUF_H
#define GPTQ_GGUF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GQ_ABI_VERSION 6

/* ggml type ids (quant_utils.py:11-16) */
enum { GQ_Q2_K = 10, GQ_Q3_K = 11, GQ_Q4_K = 12, GQ_Q5_K = 13, GQ_Q6_K = 14 };

/* element types of activations / weights / dequantized outputs */
enum { GQ_F32 = 0, GQ_F16 = 1, GQ_BF16 = 2 };

typedef enum {
    GQ_OK = 0,
    GQ_E_BAD_TYPE = -1,    /* unknown q_type / dtype */
    GQ_E_BAD_SHAPE = -2,   /* C % 256 != 0, R <= 0, ld too small, ... */
    GQ_E_WORKSPACE = -3,   /* ws == NULL or ws_bytes too small */
    GQ_E_UNSUPPORTED = -4, /* a reference option this build does not implement */
    GQ_E_HIP = -5,         /* a HIP runtime call failed (see gq_last_error) */
    GQ_E_NULL = -6
} gq_status;

/* quant_utils.py:19-26 + gguf.constants.GGML_QUANT_SIZES */
typedef struct {
    int bits;       /* 2..6 */
    int qmin, qmax; /* clamp range of the stored ints */
    int scale_maxq; /* 15 / 31 / 63 */
    int group;      /* 16 / 32 */
    int is_signed;  /* 1: int8 tensors (Q3_K, Q6_K) */
    int k_search;   /* 1: make_k_quants (Q2/Q4/Q5), 0: make_quants (Q3/Q6) */
    int type_size;  /* bytes per 256-value GGUF block: 84/110/144/176/210 */
} gq_type_info_t;

/* scale-search hyper-parameters (quant.py:94-111; python floats -> double).  ABI version 2 added the last three:
   make_quants' quant_scale (quant_utils.py:147-197, Q3_K / Q6_K only -- make_k_quants ignores it). */
typedef struct {
    double rmin;      /* -1.0 */
    double rdelta;    /*  0.1 */
    int nstep;        /*  20  */
    int quant_scale;  /* 0: "absmax" (default), 1: "mse" = the grid search of quant_utils.py:164-191, verbatim */
    int grid;         /* 100 (gptq.py:41) */
    double maxshrink; /* 0.80 (quant_utils.py:64) */
} gq_search_t;

int gq_abi_version(void);
const char* gq_last_error(void);

/* ---- environment and options -------------------------------------------------------------------------------------
   The LIBRARY reads two environment variables:
     GQ_OPTIONS    "name=value,name=value" (a bare name means 1): initial values of the options below, read once at the
                   first call; an unknown name, a value that is not an integer or one outside the option's range makes
                   EVERY entry point fail with GQ_E_UNSUPPORTED + gq_last_error (a typo must not silently measure the default)
     GQ_PROF_DUMP  file that gq_prof_collect2 appends every timed launch interval to (timeline studies)
   The PYTHON package (host code) reads:
     GQ_SO_PATH         another build of this library (kernel A/B probes)
     GQ_CHAIN_STREAMS   lanes of the block schedule (default 4 = the hardware queues the chains of a block share)
     GQ_ROW_SPLIT       0 / 1: never / always split the widest matrix by rows across ranks (default: from 4 ranks up)
     GQ_STACK           0: every Linear walks its columns alone (default 1: Linears that share a factorisation are stacked by
                        rows into one gq_gptq_quantize_stacked call; same results)
     GQ_POST_BLOCKS     early | before_last_block (default) | last: where lm_head is quantized (same tensors and files)
     GQ_FUSED_FORWARD   off | exact | all: overrides the Quantizer's fused_forward argument (A/B runs)
     GQ_SAVE_SKIP       1: no data.pth is written (measures what saving costs)
     GQ_SAVE_SLOTS, GQ_SAVE_SLOT_MB   staging slots of the data.pth writer (3 x 704 MB)
     GQ_TIMING          gpu: HIP-event split of the Quantizer's phases next to the host-side one
     GQ_TRACE           comma list of "blocks" (per-block wall time, synchronises) and "sched" (the block schedule's enqueue log)
   (quant.py also sets ROCm's GPU_MAX_HW_QUEUES=16 unless it is set.)  Nothing else is read from the environment.

   Options (gq_option_set / gq_option_get; defaults in brackets) are tuning and test switches: no option changes a result
   except where noted -- the tests flip them to prove exactly that.
     syrk_128 [0] 128x128-tile SYRK everywhere | syrk_image [0] re-laid-out operand image instead of reading X in place |
     syrk_nosplit [0] no K-split of the last round | syrk_persist [1] persistent launch with XCD rendezvous |
     syrk_wgs [0 = one per CU] resident workgroups of that launch | syrk_ck [256] half-stages (32 tokens) between the soft XCD
     rendezvous inside a tile of that launch (power of two >= 16, 0: none; results do not depend on it) | syrk_w4 [1] four waves with 128x128 wave tiles, 0 = eight
     waves with 128x64 (bit-identical) | syrk_gw [4] width in 256-tiles of the super-tile an XCD's 32 workgroups work on at a time
     (a power of two <= 32; 32 / gw rows; moves the K-split round: H within the tolerance class)
     chol_3p_min [1792] smallest half-node on the image GEMMs (0: never; changes U within the tolerance class) |
     chol_planes [2] 2 = row-scaled fp16 x 2, 3 = bf16 x 3 (tolerance class) | chol_3b_min [1024] | chol_fp32 [0] fp32 MFMA only
     (tolerance class) | chol_no_pair [0] | chol_no_equil [0] | chol_poison [0] NaN-fill scratch that must not be read |
     diag_ref [0] column-by-column leaf kernel (tolerance class)
     no_lookahead [0] | la [8] blocks per super-block | seg_pair [1] one column-loop launch per 256-column pair of blocks (0: per
     block) | near_classic [0] | near_quad [0] | near64_maxn [768] | far_sync [0] |
     far_async_max_rows [8192] | far_async_min_sb [8] | far_wgs [192] | far_bdma [1] far GEMM B operand by LDS-DMA (0: registers + ds_write) |
     chain_generic [0] | gemm32_64_max [256]
     ss_wide [-1] scale-search mapping: -1 by size, 1 eight lanes, 0 one lane, 2 a lane pair per group
     stage_host_wgs [0] workgroups of gq_stage_to_host */
int gq_option_count(void);
const char* gq_option_name(int i);                                   /* NULL past the end */
int gq_option_get(const char* name, int64_t* value);
int gq_option_default(const char* name, int64_t* value);
int gq_option_set(const char* name, int64_t value, int64_t* previous); /* previous may be NULL; takes effect for later calls */
int gq_type_info(int q_type, gq_type_info_t* out_host);

/* Scratch bytes needed by an entry point.  op: one of GQ_WS_*; unused dims = 0. */
enum { GQ_WS_H_ACCUMULATE = 1, GQ_WS_H_PREPARE = 2, GQ_WS_GPTQ_QUANTIZE = 3, GQ_WS_CHOL_GEMM = 4 /* R = M, C = N, T = K */ };
size_t gq_workspace_bytes(int op, int64_t R, int64_t C, int64_t T, int block_size);

/* replaces gptq.py:96,108-112 (GPTQ.update):  H = beta*H + alpha * X^T X.
   X[T,C] row-major in x_dtype (fp16/bf16 products are exact in fp32; accumulated
   in fp32 on the matrix cores).  H is fp32 [C,C], full square, updated in place. */
int gq_h_accumulate(float* H, const void* X, int x_dtype, int64_t T, int64_t C,
                    float beta, float alpha, void* ws, size_t ws_bytes, void* stream);

/* The same for up to 8 Hessians in ONE grid (the distinct Linear inputs of a transformer
   block: attn-in, o-in, mlp-in, down-in): H_host[i] <- beta[i]*H[i] + alpha[i]*X[i]^T X[i].
   The *_host arrays live in host memory and hold device pointers / sizes.  Workspace =
   sum of gq_workspace_bytes(GQ_WS_H_ACCUMULATE, 0, C[i], T[i], 0). */
int gq_h_accumulate_grouped(int n, float* const* H_host, const void* const* X_host, const int64_t* T_host,
                            const int64_t* C_host, const float* beta_host, const float* alpha_host,
                            int x_dtype, void* ws, size_t ws_bytes, void* stream);

/* The same with every X[i] given as nblocks[i] SEPARATE device blocks of block_tokens[i] x C[i] values each (row-major,
   16-byte aligned, block_tokens % 128 == 0, C % 256 == 0; fp16 / bf16): the activation tensors the forward hooks
   of gptq.py:79-114 receive, one per calibration sample, read where the forward left them -- no staging copy.
   blocks_host[i] is a host array of nblocks[i] device pointers.  Results are bit-identical to the same rows laid
   out contiguously.  GQ_E_UNSUPPORTED for shapes the in-place kernel does not take (use gq_h_stage then). */
int gq_h_accumulate_segments(int n, float* const* H_host, const void* const* const* blocks_host,
                             const int64_t* nblocks_host, const int64_t* block_tokens_host, const int64_t* C_host,
                             const float* beta_host, const float* alpha_host, int x_dtype, void* ws, size_t ws_bytes,
                             void* stream);

/* hook side of gptq.py:96 (`input.reshape(-1, C)`, the activations GPTQ.update consumes): appends `nbytes` of
   activation rows at `dst` inside the caller's staging buffer (device to device, any alignment).  The handle keeps
   up to 64 Ki tokens and folds them into H with ONE gq_h_accumulate -- the telescoped form of the per-sample
   updates of gptq.py:106-112. */
int gq_h_stage(void* dst, const void* src, int64_t nbytes, void* stream);
/* The staging copies of a whole fold in one launch: n blocks (host arrays of device addresses and byte counts; every
   block 16-byte aligned and a multiple of 16 bytes) are laid one behind the other at dst.  The handle of a Linear with
   ragged inputs -- an MoE expert sees a data-dependent handful of tokens per calibration sample -- keeps references to the
   hooks' tensors (as the zero-copy path does) and gathers them when the fold is due, instead of one latency-bound copy
   per sample.  ws: (2 n + 1) * 8 + 512 bytes of device scratch. */
int gq_h_stage_many(void* dst, const void* const* srcs_host, const int64_t* nbytes_host, int n, void* ws, size_t ws_bytes,
                    void* stream);

/* replaces gptq.py:134-135,141 (dead channels) + gptq.py:304-324 (_prepare) +
   linalg_utils.py:8-12: zero-column masking, damping, U = chol_upper(inv(H)).
   H and W are mutated exactly as in the reference (damping persists in H).
   *not_invertible (device int) is set to 1 and U to the identity when H is not
   positive definite (gptq.py:321-323), else 0.
   col_flags_out (device, 2*C bytes, may be NULL) receives dead[C] (H_jj == 0 on entry)
   followed by zc[C] (dead or all-zero weight column): the two sets U depends on. */
int gq_h_prepare(float* H, float* W, int64_t R, int64_t C, float rel_damp, float* U,
                 int* not_invertible, uint8_t* col_flags_out, void* ws, size_t ws_bytes, void* stream);

/* For a Linear whose Hessian is the SAME as one already prepared (q/k/v, gate/up share
   their input): applies the dead-channel set to this W (gptq.py:141) and sets *mismatch
   (device int) to 1 iff this W's zero-column set differs from col_flags' -- if 0, the
   leader's U is exactly the U the reference would compute for this Linear. */
int gq_w_prepare(const uint8_t* col_flags, float* W, int64_t R, int64_t C, int* mismatch, void* stream);

/* The payload of the path's one collective (gptq.py:131-132, all_reduce of H): H is exactly symmetric, so only
   the 128x128 tiles on and above the diagonal travel.  gq_h_pack_upper copies them into buf
   (fp32 [nt (nt+1)/2][128][128], tiles row-major over the upper triangle, nt = C/128; C % 128 == 0);
   gq_h_unpack_upper writes a reduced buf back into H and mirrors it below the diagonal.  The caller
   all-reduces buf in between (RCCL): half the bytes of reducing H itself. */
int gq_h_pack_upper(const float* H, int64_t C, float* buf, void* stream);
int gq_h_unpack_upper(const float* buf, int64_t C, float* H, void* stream);

/* replaces quant_utils.py:90-145 (Quantizer.get_scale_and_zero) incl.
   make_k_quants :199-274 / make_quants :147-197, on one [rows,256] panel with row
   stride ld (elements).  d/dmin element r at d[r*d_stride]; s/m row r at s + r*s_ld. */
int gq_scale_search(const float* x, int64_t rows, int64_t ld, int q_type, const gq_search_t* p_host,
                    uint16_t* d, int64_t d_stride, uint8_t* s, int64_t s_ld,
                    uint16_t* dmin, int64_t dmin_stride, uint8_t* m, int64_t m_ld, void* stream);

/* The same panel search, additionally returning what make_k_quants / make_quants return
   (quant_utils.py:147-197, :199-274): the per-group fp32 scale and zero, [rows, 256/G] each.
   x may be fp32, or fp16/bf16 (then every op rounds to that dtype like the reference's RTN of
   embed/lm_head, quantizer.py:109,195).  d/dmin are [rows], s/m [rows, 256/G], all contiguous. */
int gq_group_search(const void* x, int x_dtype, int64_t rows, int64_t ld, int q_type, const gq_search_t* p_host,
                    float* group_scale, float* group_zero, uint16_t* d, uint8_t* s, uint16_t* dmin, uint8_t* m,
                    void* stream);

/* replaces gptq.py:145-276 (the rank-0 body of GPTQ.step): blocked column-wise
   quantize + error feedback + trailing update.  W (fp32 working copy) becomes the
   dequantized matrix; U is chol_upper(H^-1).  block_size <= 0 means C.
   Requires block_size % 16 == 0.  act_order: gq_gptq_quantize_perm below. */
int gq_gptq_quantize(float* W, const float* U, int64_t R, int64_t C, int q_type,
                     int block_size, int static_groups, const gq_search_t* p_host,
                     uint8_t* qweight, uint16_t* d, uint8_t* s, uint16_t* dmin, uint8_t* m,
                     void* ws, size_t ws_bytes, void* stream);

/* The same call on SEVERAL Linears that share U -- the Linears of one input tensor whose dead / zero-column sets agree:
   q / k / v, gate / up, an expert's w1 / w3 (gptq.py:304-324 gives them the same H and therefore the same U) -- stacked
   by rows into one [R, C] working copy: matrix k is rows [row_ends[k-1], row_ends[k]) (host array of n_stacked ascending
   multiples of 64, the last one == R; n_stacked <= 8).  Rows never mix in gptq.py:222-270, and the column loop is bound
   by its C dependent steps, not by its rows: one walk over the columns instead of n_stacked.  The one place where the
   reference looks across the rows of a matrix -- the panel-wide `continue` of quant_utils.py:250-252 inside
   get_scale_and_zero -- is evaluated per stacked matrix, so every output row is bit-identical to the separate calls. */
int gq_gptq_quantize_stacked(float* W, const float* U, int64_t R, int64_t C, int q_type,
                             int block_size, int static_groups, const gq_search_t* p_host,
                             uint8_t* qweight, uint16_t* d, uint8_t* s, uint16_t* dmin, uint8_t* m,
                             const int64_t* row_ends_host, int n_stacked,
                             void* ws, size_t ws_bytes, void* stream);

/* gq_gptq_quantize on a ROW SLICE of a matrix that several ranks quantize together (rows are independent in
   gptq.py:222-270 given U; SURVEY section 8(e): from 4 ranks up the widest Linear of a block is cut R/k rows per GPU).
   The one place where the reference looks across ALL rows of a matrix is make_k_quants' `if not valid.any(): continue`
   (quant_utils.py:250-252): a slice evaluates it over its own rows.  The slice's results equal the whole matrix's rows
   bit for bit unless some slice had to search a panel AGAIN because of that test (the slow path of the scale search:
   a panel with an iteration in which NO group is valid and whose candidate some group had taken).
   *panel_researches (device int32, written on `stream`) = the number of such re-searches in this call.  The caller sums
   the ranks' counts (they travel with the block's all-gather) and, when the sum is not 0, quantizes that matrix whole
   instead (gq_gptq_quantize on all rows, same U): always the N = 1 bytes.  0 for Q3_K / Q6_K (no search). */
int gq_gptq_quantize_slice(float* W, const float* U, int64_t R, int64_t C, int q_type,
                           int block_size, int static_groups, const gq_search_t* p_host,
                           uint8_t* qweight, uint16_t* d, uint8_t* s, uint16_t* dmin, uint8_t* m,
                           int32_t* panel_researches, void* ws, size_t ws_bytes, void* stream);

/* 1 if gq_gptq_quantize / gq_gptq_quantize_perm / gq_obq_quantize on an R x C matrix MAY put the bulk of its far
   trailing updates (gptq.py:270 beyond the current 1024-column super-block) on the library's own helper HIP stream,
   next to the column loop on the caller's stream (few rows, many super-blocks: the loop's kernels are latency-bound
   and leave most of the chip idle); one call per device holds the helper at a time, a call that finds it taken runs
   on the caller's stream alone (same results).  All ordering is by events; the call still completes in stream order for the
   caller.  A scheduler that runs several chains on streams of its own uses one stream fewer then: more than four
   hardware queues cost more than the overlap gains.  GQ_FAR_SYNC=1 switches the helper off. */
int gq_gptq_uses_helper_stream(int64_t R, int64_t C, int block_size);
/* Process-wide switch of that helper stream for the calls enqueued from now on (1 = allowed, the default); returns the
   previous setting.  A scheduler with more chains than streams keeps all its streams and switches the helper off. */
int gq_far_helper_enable(int on);

/* GPTQ.step with act_order=True (gptq.py:208-216, 233-235, 272-276; implies static_groups, gptq.py:45-46;
   not for Q3_K, gptq.py:204-206).  The caller permutes: perm = argsort(diag(H), descending) (int32 [C], on the
   device), W <- W[:, perm], H <- H[perm][:, perm], U = gq_h_prepare(H, W).  d/s/dmin/m are INPUTS: the static
   scales of the ORIGINAL column groups (gq_scale_search on the unpermuted W, gptq.py:184-196); column j is
   quantized with the parameters of group perm[j]/G of super-group perm[j]/256.  qweight comes back in
   permuted positions (the caller applies argsort(perm)); W becomes the dequantized matrix, permuted. */
int gq_gptq_quantize_perm(float* W, const float* U, int64_t R, int64_t C, int q_type, int block_size,
                          const int32_t* perm, const uint16_t* d, const uint8_t* s, const uint16_t* dmin,
                          const uint8_t* m, uint8_t* qweight, void* ws, size_t ws_bytes, void* stream);

/* ---- EvoPress' FastOBQ (evopress/src/fast_obq.py): the same column loop on uniform min/max grids ----
   replaces fast_obq.py:133-141 (dead diagonal, damping, W[:, dead] = 0) + :219-232 (_prepare: zero-column mask,
   U = chol_upper(inv(H))): as gq_h_prepare, but damping comes BEFORE the mask and masked diagonals stay exactly 1.
   A non-positive-definite H sets *not_invertible (the reference raises from torch.linalg.cholesky). */
int gq_obq_h_prepare(float* H, float* W, int64_t R, int64_t C, float rel_damp, float* U, int* not_invertible,
                     uint8_t* col_flags_out, void* ws, size_t ws_bytes, void* stream);

/* replaces fast_obq.py:146-200 for ONE bit width (1..8) given U: q = clamp(round(w / max(scale, 1e-9) + zero), 0,
   2^bits - 1), w_hat = scale (q - zero) (quant_utils.py:23-29), the grid of a group found from W when its first
   column is reached (quant_utils.py:57-106, fast_obq.py:168-171).  group_size 0: one grid per row from the original
   W (:153-154; returned in scale/zero [R, 1], which the reference leaves uninitialised).  group_size % 16 == 0,
   C % group_size == 0, C % 256 == 0, block_size % 16 == 0.  qweight u8 [R, C]; scale / zero fp32 [R, C/group_size];
   W becomes the dequantized matrix.  Workspace: GQ_WS_GPTQ_QUANTIZE. */
int gq_obq_quantize(float* W, const float* U, int64_t R, int64_t C, int bits, int group_size, int sym, int block_size,
                    uint8_t* qweight, float* scale, float* zero, void* ws, size_t ws_bytes, void* stream);

/* replaces quantizer.py:278-330 (_quant_non_block_module, RTN for embed/lm_head).
   W in w_dtype; GQ_F32 follows the fp32 arithmetic of the reference run with
   --dtype float32. */
int gq_rtn_quantize(const void* W, int w_dtype, int64_t R, int64_t C, int q_type,
                    const gq_search_t* p_host,
                    uint8_t* qweight, uint16_t* d, uint8_t* s, uint16_t* dmin, uint8_t* m,
                    void* stream);

/* replaces quant_utils.py:277-310 (dequantize_linear_weight) + the caller's cast
   to the model dtype (quantizer.py:264). */
int gq_dequantize(int q_type, const uint8_t* qweight, const uint16_t* d, const uint8_t* s,
                  const uint16_t* dmin, const uint8_t* m, int64_t R, int64_t C,
                  void* out, int out_dtype, void* stream);

/* replaces packing_utils.py:33-326 (pack_Q2K .. pack_Q6K, pack_scale_min_torch).
   Inputs are const: the +4/+32 offsets of Q3_K/Q6_K are applied on the fly.
   dmin/m may be NULL for Q3_K/Q6_K.  out: [R, C/256*type_size] bytes. */
int gq_pack(int q_type, const uint8_t* qweight, const uint16_t* d, const uint8_t* s,
            const uint16_t* dmin, const uint8_t* m, int64_t R, int64_t C, uint8_t* out, void* stream);

/* C[M,N] (ldc) -= A[M,K] (lda) @ B[K,N] (ldb), fp32, each output a k-ordered fma
   chain from 0 followed by one subtraction: the trailing update of gptq.py:270,
   exposed for tests and benchmarks. */
int gq_trailing_update(float* Cmat, int64_t ldc, const float* A, int64_t lda, const float* B, int64_t ldb,
                       int64_t M, int64_t N, int64_t K, void* stream);

/* One product of gq_h_prepare's blocked Cholesky / triangular inverse (linalg_utils.py:8-12) through the kernels its top
   recursion levels use, exposed for tests and benchmarks: every fp32 operand is split once into 16-bit planes
   (planes = 3: bf16, exact split, six products per term; planes = 2: row-scaled fp16, three products) and multiplied on
   the 16-bit matrix cores with fp32 accumulation -- fp32-GEMM accuracy, NOT the bit-exact chain of gq_trailing_update.
     mode 0: C -= A op(B);  1: C = A op(B);  2: C = -(A op(B));   A is [M,K];  op(B) = B^T with B [N,K] if trans_b, else B [K,N];
     k_range 0: all k;  1: k < 256 (tile col + 1) (B^T lower-triangular; trans_b);  2: k >= 256 tile col (B lower-triangular;
     !trans_b);  3: k < 256 (tile row + 1) (A lower-triangular);  lower: only 256-tiles with tile row >= tile col.
   128 x 128 blocks of a triangular operand beyond its block diagonal are never read.  M, N % 256 == 0, K % 128 == 0.
   Workspace: gq_workspace_bytes(GQ_WS_CHOL_GEMM, M, N, K, 0). */
int gq_chol_gemm(float* Cmat, int64_t ldc, const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M, int64_t N,
                 int64_t K, int trans_b, int mode, int k_range, int lower, int planes, void* ws, size_t ws_bytes, void* stream);

/* Device -> pinned host copy by a kernel on `stream` (the data.pth staging slots of quantizer.py's saver; reference
   quantizer.py:267-275 moves the tensors with .cpu()).  host_dst must be pinned and mapped (hipHostRegister /
   hipHostMalloc).  Unlike hipMemcpyAsync it takes no runtime copy-engine lock: a side thread staging results does not
   stall the launches of the thread that runs the forward. */
int gq_stage_to_host(void* host_dst, const void* src, int64_t nbytes, void* stream);

/* SURVEY 8(f) row 2 (Quantizer(fused_forward="exact" | "all")): the elementwise part of the calibration forward the
   reference leaves to HF eager modules (quantizer.py:293 `block(inp_batch, **kwargs)`), one HBM pass each, fp32
   arithmetic rounded to `dtype` (GQ_F16 / GQ_BF16) after every torch op of the module it replaces:
     gq_fwd_rmsnorm   LlamaRMSNorm.forward:  out = weight * dtype(float(x) * rsqrt(mean(float(x)^2) + eps));  x [tokens, C], C % 8 == 0
     gq_fwd_rope      apply_rotary_pos_emb on one of q / k:  out = dtype(x cos) + dtype(rotate_half(x) sin);
                      x, out [tokens, heads, head_dim] (the projection's own layout), cos / sin [tokens, head_dim], head_dim % 16 == 0
     gq_fwd_silu_mul  LlamaMLP.forward's act_fn(gate) * up over n elements, n % 8 == 0
   All pointers 16-byte aligned, tensors contiguous, out does not overlap an input. */
int gq_fwd_rmsnorm(const void* x, const void* weight, void* out, int64_t tokens, int64_t C, float eps, int dtype, void* stream);
/* gq_fwd_rmsnorm with mean(x^2) summed in the order of ATen's reduce kernel on a 64-wide wavefront (C % 512 == 0): bit-identical
   to the eager module on the PyTorch this was written against; the Python host verifies that on first use per (C, dtype)
   and keeps the eager module otherwise.  stats (optional, fp32 [tokens, 2]): the row's mean(x^2) and rsqrt(mean + eps) as
   the kernel computed them -- what that verification compares with torch's own, bit for bit. */
int gq_fwd_rmsnorm_ordered(const void* x, const void* weight, void* out, int64_t tokens, int64_t C, float eps, int dtype,
                           float* stats, void* stream);
int gq_fwd_rope(const void* x, const void* cos_, const void* sin_, void* out, int64_t tokens, int heads, int head_dim, int dtype,
                void* stream);
int gq_fwd_silu_mul(const void* gate, const void* up, void* out, int64_t n, int dtype, void* stream);

/* Optional HIP-event timing of the library's own kernels, on the stream they are
   launched on (bench.py's roofline leg).  tag_mask bit t enables tag t; collect()
   synchronises the recorded events and ADDS elapsed ms / launch counts per tag. */
void gq_prof_enable(unsigned tag_mask);
int gq_prof_ntags(void);
const char* gq_prof_name(int tag);
int gq_prof_collect(double* ms_host, long* n_host);
/* as gq_prof_collect; busy_ms[tag] += length of the union of the tag's launch intervals (launches of one
 * tag overlapping on different streams count once) */
int gq_prof_collect2(double* ms_host, long* n_host, double* busy_ms_host);

#ifdef __cplusplus
}
#endif
#endif
