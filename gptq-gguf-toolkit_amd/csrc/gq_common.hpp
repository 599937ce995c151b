// gq_common.hpp -- shared host/device helpers for libgptqgguf_hip.so (gfx950 only).
//
// Build flags that matter for results (see csrc/Makefile):
//   -ffp-contract=off   every a*b+c in the codec kernels is TWO roundings unless it is
//                       spelled fmaf(); the reference (ATen CPU) does not contract.
//   no -ffast-math; hipcc's default correctly rounded fp32 divide/sqrt and f32
//   denormals stay on.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gptq_gguf.h"

namespace gq {

extern thread_local char g_err[512];

#define GQ_FAIL(code, ...)                                \
    do {                                                  \
        snprintf(gq::g_err, sizeof(gq::g_err), __VA_ARGS__); \
        return (code);                                    \
    } while (0)

// (a failed call leaves the runtime's sticky "last error" set: it is consumed here, so that the NEXT launch check of the
// process does not report it again for an unrelated kernel)
#define GQ_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            (void)hipGetLastError();                                                         \
            GQ_FAIL(GQ_E_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));                \
        }                                                                                    \
    } while (0)

#define GQ_LAUNCH_CHECK() GQ_HIP(hipGetLastError())

// ---- library options: every tuning / test switch of the library in ONE table (include/gptq_gguf.h documents them).
// Set through gq_option_set() or, once at load, from the single environment variable
//   GQ_OPTIONS="name=value,name=value"
// and read per call with opt() (a relaxed atomic load).  X(name, default, smallest, largest accepted value).
#define GQ_OPTION_LIST(X)                                                                                                         \
    /* K1 SYRK */                                                                                                                 \
    X(syrk_128, 0, 0, 1)          /* 1: the 128x128-tile kernel for every shape */                                                \
    X(syrk_image, 0, 0, 1)        /* 1: re-laid-out operand image + syrk16_256e_kernel instead of reading X in place */           \
    X(syrk_nosplit, 0, 0, 1)      /* 1: no K-split of the last, partial round of tiles */                                         \
    X(syrk_persist, 1, 0, 1)      /* 0: one tile per workgroup instead of the persistent launch with XCD rendezvous */            \
    X(syrk_wgs, 0, 0, 4096)          /* resident workgroups of the persistent launch (0: one per CU) */                           \
    X(syrk_ck, 256, 0, 65536)     /* half-stages between the soft XCD rendezvous inside a tile (a power of two >= 16; 0: none) */  \
    X(syrk_gw, 4, 1, 32)          /* width in tiles of the super-tile an XCD's 32 workgroups share (1, 2, 4, 8, 16, 32; 32 / gw rows) */ \
    X(syrk_w4, 1, 0, 1)           /* 0: eight waves with 128x64 wave tiles (syrk16_256n_kernel) instead of four with 128x128 */   \
    /* K3 Cholesky chain */                                                                                                       \
    X(chol_3p_min, 1792, 0, 1048576)    /* smallest half of a recursion node that runs on the image GEMMs (0: never) */           \
    X(chol_planes, 2, 2, 3)       /* 2: row-scaled fp16 x 2 images, 3: exact bf16 x 3 */                                          \
    X(chol_3b_min, 1024, 0, 1048576)    /* smallest half that runs on the on-the-fly split-bf16 GEMM */                           \
    X(chol_fp32, 0, 0, 1)         /* 1: v_mfma_f32_32x32x2_f32 everywhere below the image levels */                               \
    X(chol_no_pair, 0, 0, 1)      /* 1: SYRK update and L21 X11 of a small node as two launches */                                \
    X(chol_no_equil, 0, 0, 1)     /* 1: no power-of-two equilibration */                                                          \
    X(chol_poison, 0, 0, 1)       /* 1: NaN-fill the scratch the chain must never read (tests) */                                 \
    X(diag_ref, 0, 0, 1)          /* 1: the column-by-column 128x128 leaf kernel (reference for tests) */                         \
    /* K5/K6 column loop */                                                                                                       \
    X(no_lookahead, 0, 0, 1)      /* 1: trailing update after every block, no chained far update */                               \
    X(la, 8, 2, 8)                /* blocks per look-ahead super-block (even, 2..8) */                                            \
    X(seg_pair, 1, 0, 1)          /* 0: one column-loop launch per 128-column block instead of one per 256-column pair */          \
    X(near_classic, 0, 0, 1)      /* 1: a near launch after every block instead of the pair form */                               \
    X(near_quad, 0, 0, 1)         /* 1: near launches after every second pair */                                                  \
    X(near64_maxn, 768, 0, 1048576)     /* widest near update that takes gemm32_near256_kernel */                                 \
    X(far_sync, 0, 0, 1)          /* 1: far updates on the caller's stream (no helper stream) */                                  \
    X(far_async_max_rows, 8192, 0, 1073741824) X(far_async_min_sb, 8, 0, 1048576) /* shape window of the helper-stream form */    \
    X(far_wgs, 192, 1, 4096)         /* resident workgroups of the helper's persistent far GEMM */                                \
    X(far_bdma, 1, 0, 1)          /* 0: the far GEMM B operand through registers + ds_write instead of LDS-DMA */                 \
    X(chain_generic, 0, 0, 1)     /* 1: the generic chained kernel instead of the dedicated far kernel */                         \
    X(gemm32_64_max, 256, 0, 1073741824)   /* problems with fewer 128-tiles than this take 64x64 tiles (0: never) */              \
    /* K4 */                                                                                                                      \
    X(ss_wide, -1, -1, 2)          /* scale-search mapping: -1 by size, 1 eight lanes, 0 one lane, 2 a lane pair per group */     \
    /* saver */                                                                                                                   \
    X(stage_host_wgs, 0, 0, 65536)    /* workgroups of gq_stage_to_host (0: default) */

enum Opt {
#define GQ_X(name, def, lo, hi) OPT_##name,
    GQ_OPTION_LIST(GQ_X)
#undef GQ_X
    OPT_COUNT
};
int64_t opt(Opt o);
int options_ok();  // GQ_OK, or GQ_E_UNSUPPORTED + gq_last_error when GQ_OPTIONS did not parse (checked by every entry point)

constexpr int GQ_MAX_STACK = 8;  // row-stacked matrices of one gq_gptq_quantize_stacked call
// The panel block of a scale-search call chain (256 bytes): words [2k], [2k + 1] = "some group valid" / "some group took the
// candidate" bits per search iteration of stacked matrix k (left at zero by every launch); word GQ_PANEL_RESEARCH counts the
// panel-wide re-searches of the chain (panel_fixup_kernel; read by gq_gptq_quantize_slice)
constexpr int GQ_PANEL_RESEARCH = 2 * GQ_MAX_STACK;

struct TypeInfo {
    int bits, qmin, qmax, scale_maxq, group, is_signed, k_search, type_size;
};

// reference quant_utils.py:19-26
inline bool type_info(int q, TypeInfo& t) {
    switch (q) {
    case GQ_Q2_K: t = {2, 0, 3, 15, 16, 0, 1, 84}; return true;
    case GQ_Q3_K: t = {3, -4, 3, 31, 16, 1, 0, 110}; return true;
    case GQ_Q4_K: t = {4, 0, 15, 63, 32, 0, 1, 144}; return true;
    case GQ_Q5_K: t = {5, 0, 31, 63, 32, 0, 1, 176}; return true;
    case GQ_Q6_K: t = {6, -32, 31, 63, 16, 1, 0, 210}; return true;
    default: return false;
    }
}

__device__ __forceinline__ float h2f(uint16_t h) {
    _Float16 v = __builtin_bit_cast(_Float16, h);
    return (float)v;
}
__device__ __forceinline__ uint16_t f2h(float f) {  // RNE, denormals kept
    // The empty asm pins `f` as an fp32 value in a VGPR.  Without it the backend folds
    // fptrunc(fmul/fma f32) into v_fma_mixlo_f16, which rounds the EXACT product once to fp16; the
    // reference (ATen) rounds to fp32 first and then to fp16 (measured: 1 group in ~4600 of an fp16
    // weight picked a different scale-search candidate through such a double-rounding tie).
    asm volatile("" : "+v"(f));
    _Float16 v = (_Float16)f;
    return __builtin_bit_cast(uint16_t, v);
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {  // RNE
    uint32_t x = __builtin_bit_cast(uint32_t, f);
    if ((x & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((x >> 16) | 0x40);
    x += 0x7fffu + ((x >> 16) & 1);
    return (uint16_t)(x >> 16);
}

// torch.clamp(v, lo, hi) for NaN-free v
__device__ __forceinline__ float clampf(float v, float lo, float hi) {
    v = v < lo ? lo : v;
    v = v > hi ? hi : v;
    return v;
}

__device__ __forceinline__ float ival(uint8_t b, int is_signed) {
    return is_signed ? (float)(int8_t)b : (float)b;
}

// reference quant_utils.py:34-40 / :43-46 with ds = f32(d)*s, dm = f32(dmin)*m precomputed
__device__ __forceinline__ float quantize1(float x, float ds, float dm, float qmin, float qmax) {
    float den = ds < 1e-9f ? 1e-9f : ds;
    return clampf(rintf((x + dm) / den), qmin, qmax);
}
__device__ __forceinline__ float dequantize1(float q, float ds, float dm) { return ds * q - dm; }

// ---- optional per-kernel timing with HIP events on the launch stream (bench.py) ----
enum ProfTag {
    PT_TRANSPOSE = 0, PT_SYRK, PT_PREP_ELEM, PT_DIAG_POTRF, PT_CHOL_GEMM, PT_TRTRI_GEMM, PT_SCALE_SEARCH,
    PT_GPTQ_SEGMENT, PT_TRAILING, PT_BLOCK_FAR, PT_DEQUANT, PT_RTN, PT_PACK, PT_TRAILING_FAR, PT_CHOL_IMG_GEMM,
    PT_CHOL_SPLIT, PT_COUNT
};
extern unsigned g_prof_mask;
void prof_begin(int tag, hipStream_t st);
void prof_end(int tag, hipStream_t st);
struct ProfScope {
    int tag;
    hipStream_t st;
    bool on;
    ProfScope(int t, hipStream_t s) : tag(t), st(s), on((g_prof_mask >> t) & 1u) {
        if (on) prof_begin(tag, st);
    }
    ~ProfScope() {
        if (on) prof_end(tag, st);
    }
};

}  // namespace gq
