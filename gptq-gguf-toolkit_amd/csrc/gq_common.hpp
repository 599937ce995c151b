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

#define GQ_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) GQ_FAIL(GQ_E_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

#define GQ_LAUNCH_CHECK() GQ_HIP(hipGetLastError())

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
