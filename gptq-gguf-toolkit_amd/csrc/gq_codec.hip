// gq_codec.hip -- K7 (dequantize), K8 elementwise half (RTN quantize), K9-K13 (bit-packers).
//
// All HBM-bound byte/elementwise work: one pass over the inputs, 16-byte accesses.
//   dequantize : 1 B/param in (+ ~0.2 B of scales)  -> 2 or 4 B/param out
//   pack       : 1 B/param in                        -> type_size/256 B/param out
#include "gq_common.hpp"

namespace gq {

// ------------------------------------------------------------------ dequantize
// reference quant_utils.py:277-310; one thread = 16 consecutive values (never
// straddles a group: G is 16 or 32).
template <typename OutT>
__device__ __forceinline__ OutT cvt_out(float v);
template <>
__device__ __forceinline__ float cvt_out<float>(float v) { return v; }
struct half_bits { uint16_t b; };
struct bf16_bits { uint16_t b; };
template <>
__device__ __forceinline__ half_bits cvt_out<half_bits>(float v) { return {f2h(v)}; }
template <>
__device__ __forceinline__ bf16_bits cvt_out<bf16_bits>(float v) { return {f2bf(v)}; }

template <typename OutT>
__global__ __launch_bounds__(256) void dequantize_kernel(
    const uint8_t* __restrict__ q, const uint16_t* __restrict__ d, const uint8_t* __restrict__ s,
    const uint16_t* __restrict__ dmin, const uint8_t* __restrict__ m, int64_t R, int64_t C, int G,
    int is_signed, OutT* __restrict__ out) {
    const int64_t n16 = R * C / 16;
    const int64_t per_row = C / 16;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / per_row, c = (i % per_row) * 16;
        const int64_t sg = r * (C / 256) + c / 256, g = r * (C / G) + c / G;
        const float ds = h2f(d[sg]) * ival(s[g], is_signed);
        const float dm = h2f(dmin[sg]) * ival(m[g], is_signed);
        const uint4 qv = *reinterpret_cast<const uint4*>(q + r * C + c);
        const uint32_t qw[4] = {qv.x, qv.y, qv.z, qv.w};
        OutT o[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            uint8_t b = (uint8_t)(qw[k >> 2] >> (8 * (k & 3)));
            o[k] = cvt_out<OutT>(dequantize1(ival(b, is_signed), ds, dm));
        }
        OutT* op = out + r * C + c;
        if constexpr (sizeof(OutT) == 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) reinterpret_cast<uint4*>(op)[k] = reinterpret_cast<const uint4*>(o)[k];
        } else {
#pragma unroll
            for (int k = 0; k < 2; ++k) reinterpret_cast<uint4*>(op)[k] = reinterpret_cast<const uint4*>(o)[k];
        }
    }
}

int launch_dequantize(int q_type, const uint8_t* q, const uint16_t* d, const uint8_t* s, const uint16_t* dmin,
                      const uint8_t* m, int64_t R, int64_t C, void* out, int out_dtype, hipStream_t st) {
    TypeInfo ti;
    if (!type_info(q_type, ti)) GQ_FAIL(GQ_E_BAD_TYPE, "gq_dequantize: unknown q_type %d", q_type);
    if (R <= 0 || C <= 0 || C % 256) GQ_FAIL(GQ_E_BAD_SHAPE, "gq_dequantize: R=%ld C=%ld", (long)R, (long)C);
    const int64_t n16 = R * C / 16;
    dim3 grid((unsigned)((n16 + 255) / 256 < 8192 ? (n16 + 255) / 256 : 8192)), block(256);
    ProfScope ps(PT_DEQUANT, st);
    switch (out_dtype) {
    case GQ_F32:
        hipLaunchKernelGGL(dequantize_kernel<float>, grid, block, 0, st, q, d, s, dmin, m, R, C, ti.group,
                           ti.is_signed, (float*)out);
        break;
    case GQ_F16:
        hipLaunchKernelGGL(dequantize_kernel<half_bits>, grid, block, 0, st, q, d, s, dmin, m, R, C, ti.group,
                           ti.is_signed, (half_bits*)out);
        break;
    case GQ_BF16:
        hipLaunchKernelGGL(dequantize_kernel<bf16_bits>, grid, block, 0, st, q, d, s, dmin, m, R, C, ti.group,
                           ti.is_signed, (bf16_bits*)out);
        break;
    default: GQ_FAIL(GQ_E_BAD_TYPE, "gq_dequantize: unknown out_dtype %d", out_dtype);
    }
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

// ------------------------------------------------------------- RTN quantize
// reference quantizer.py:318-330: one vectorised quantize() over the unmodified
// weight with the per-group parameters expanded.  One thread = 16 values.
template <int WDT>
__device__ __forceinline__ float load_w(const void* W, int64_t idx) {
    if constexpr (WDT == GQ_F32) return reinterpret_cast<const float*>(W)[idx];
    else if constexpr (WDT == GQ_F16) return h2f(reinterpret_cast<const uint16_t*>(W)[idx]);
    else return bf2f(reinterpret_cast<const uint16_t*>(W)[idx]);
}

template <int WDT>
__global__ __launch_bounds__(256) void rtn_quantize_kernel(
    const void* __restrict__ W, const uint16_t* __restrict__ d, const uint8_t* __restrict__ s,
    const uint16_t* __restrict__ dmin, const uint8_t* __restrict__ m, int64_t R, int64_t C, int G, int is_signed,
    float qmin, float qmax, uint8_t* __restrict__ q) {
    const int64_t n16 = R * C / 16, per_row = C / 16;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / per_row, c = (i % per_row) * 16;
        const int64_t sg = r * (C / 256) + c / 256, g = r * (C / G) + c / G;
        const float ds = h2f(d[sg]) * ival(s[g], is_signed);
        const float dm = h2f(dmin[sg]) * ival(m[g], is_signed);
        uint32_t o[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float v = quantize1(load_w<WDT>(W, r * C + c + k), ds, dm, qmin, qmax);
            uint8_t b = is_signed ? (uint8_t)(int8_t)v : (uint8_t)v;
            o[k >> 2] |= (uint32_t)b << (8 * (k & 3));
        }
        *reinterpret_cast<uint4*>(q + r * C + c) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

int launch_rtn_elementwise(const void* W, int w_dtype, const uint16_t* d, const uint8_t* s, const uint16_t* dmin,
                           const uint8_t* m, int64_t R, int64_t C, const TypeInfo& ti, uint8_t* q,
                           hipStream_t st) {
    const int64_t n16 = R * C / 16;
    dim3 grid((unsigned)((n16 + 255) / 256 < 8192 ? (n16 + 255) / 256 : 8192)), block(256);
    const float qmin = (float)ti.qmin, qmax = (float)ti.qmax;
    ProfScope ps(PT_RTN, st);
    switch (w_dtype) {
    case GQ_F32:
        hipLaunchKernelGGL(rtn_quantize_kernel<GQ_F32>, grid, block, 0, st, W, d, s, dmin, m, R, C, ti.group,
                           ti.is_signed, qmin, qmax, q);
        break;
    case GQ_F16:
        hipLaunchKernelGGL(rtn_quantize_kernel<GQ_F16>, grid, block, 0, st, W, d, s, dmin, m, R, C, ti.group,
                           ti.is_signed, qmin, qmax, q);
        break;
    case GQ_BF16:
        hipLaunchKernelGGL(rtn_quantize_kernel<GQ_BF16>, grid, block, 0, st, W, d, s, dmin, m, R, C, ti.group,
                           ti.is_signed, qmin, qmax, q);
        break;
    default: GQ_FAIL(GQ_E_BAD_TYPE, "gq_rtn_quantize: unknown w_dtype %d", w_dtype);
    }
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

// ---------------------------------------------------------------- bit-packers
// reference packing_utils.py:8-326.  A workgroup packs PB = 32 consecutive
// 256-value blocks per turn: the 32*256 input bytes are staged in LDS with coalesced
// 16-byte loads (two per thread in flight), every thread assembles FOUR consecutive
// output bytes from LDS and stores them as one dword, and the 32*type_size output
// bytes (a multiple of 16 for every type) leave as coalesced 16-byte stores.  (r01: 8
// blocks and byte-wide LDS stores per turn kept too few bytes in flight, 2.0-2.3 TB/s.)
constexpr int PB = 32;

__device__ __forceinline__ uint8_t pack_scale_min_byte(const uint8_t* sc, const uint8_t* mn, int j) {
    // packing_utils.py:8-30, byte j of the 12
    if (j < 4) return (uint8_t)(sc[j] | ((sc[4 + j] >> 4) << 6));
    if (j < 8) return (uint8_t)(mn[j - 4] | ((mn[j] >> 4) << 6));
    return (uint8_t)((sc[j - 4] & 0x0F) | ((mn[j - 4] & 0x0F) << 4));
}

template <int QT>
__device__ __forceinline__ uint8_t pack_byte(const uint8_t* q, const uint8_t* sb, const uint8_t* mb, uint16_t d,
                                             uint16_t dmin, int o) {
    if constexpr (QT == GQ_Q2_K) {  // :33-77  scales[16] qs[64] d dmin
        if (o < 16) return (uint8_t)((sb[o] & 0x0F) | ((mb[o] & 0x0F) << 4));
        if (o < 80) {
            int t = o - 16, ch = t >> 5, l = t & 31;
            const uint8_t* c = q + ch * 128;
            return (uint8_t)(c[l] | (c[32 + l] << 2) | (c[64 + l] << 4) | (c[96 + l] << 6));
        }
        if (o < 82) return (uint8_t)(d >> (8 * (o - 80)));
        return (uint8_t)(dmin >> (8 * (o - 82)));
    } else if constexpr (QT == GQ_Q3_K) {  // :80-142  hmask[32] qs[64] scales[12] d
        if (o < 32) {
            uint8_t h = 0;
#pragma unroll
            for (int b = 0; b < 8; ++b) h |= (uint8_t)(((uint8_t)((int8_t)q[b * 32 + o] + 4) > 3) << b);
            return h;
        }
        if (o < 96) {
            int t = o - 32, ch = t >> 5, l = t & 31;
            const uint8_t* c = q + ch * 128;
            uint8_t v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint8_t u = (uint8_t)((int8_t)c[32 * k + l] + 4);
                v[k] = u > 3 ? (uint8_t)(u - 4) : u;
            }
            return (uint8_t)(v[0] | (v[1] << 2) | (v[2] << 4) | (v[3] << 6));
        }
        if (o < 108) {
            int j = o - 96;  // :103-115
            auto sc = [&](int k) { return (uint8_t)((int8_t)sb[k] + 32); };
            if (j < 8) return (uint8_t)((sc(j) & 0x0F) | ((sc(j + 8) & 0x0F) << 4));
            int t = j - 8;
            return (uint8_t)(((sc(t) >> 4) & 3) | (((sc(t + 4) >> 4) & 3) << 2) | (((sc(t + 8) >> 4) & 3) << 4) |
                             (((sc(t + 12) >> 4) & 3) << 6));
        }
        return (uint8_t)(d >> (8 * (o - 108)));
    } else if constexpr (QT == GQ_Q4_K) {  // :145-190  d dmin scales[12] qs[128]
        if (o < 2) return (uint8_t)(d >> (8 * o));
        if (o < 4) return (uint8_t)(dmin >> (8 * (o - 2)));
        if (o < 16) return pack_scale_min_byte(sb, mb, o - 4);
        int t = o - 16, base = (t >> 5) * 64, l = t & 31;
        return (uint8_t)(q[base + l] | (q[base + 32 + l] << 4));
    } else if constexpr (QT == GQ_Q5_K) {  // :193-262  d dmin scales[12] qh[32] ql[128]
        if (o < 2) return (uint8_t)(d >> (8 * o));
        if (o < 4) return (uint8_t)(dmin >> (8 * (o - 2)));
        if (o < 16) return pack_scale_min_byte(sb, mb, o - 4);
        if (o < 48) {
            int j = o - 16;
            uint8_t h = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                h |= (uint8_t)((q[64 * k + j] > 15) << (2 * k));
                h |= (uint8_t)((q[64 * k + 32 + j] > 15) << (2 * k + 1));
            }
            return h;
        }
        int t = o - 48, base = (t >> 5) * 64, j = t & 31;
        return (uint8_t)((q[base + j] & 15) | ((q[base + 32 + j] & 15) << 4));
    } else {  // Q6_K :265-326  ql[128] qh[64] scales[16] d
        auto v = [&](int idx) { return (uint8_t)((int8_t)q[idx] + 32); };
        if (o < 128) {
            int ch = o >> 6, t = o & 63, l = t & 31, hi = t >> 5;
            int b = ch * 128 + l + 32 * hi;
            return (uint8_t)((v(b) & 0xF) | ((v(b + 64) & 0xF) << 4));
        }
        if (o < 192) {
            int t = o - 128, ch = t >> 5, l = t & 31, b = ch * 128 + l;
            return (uint8_t)(((v(b) >> 4) & 3) | (((v(b + 32) >> 4) & 3) << 2) | (((v(b + 64) >> 4) & 3) << 4) |
                             (((v(b + 96) >> 4) & 3) << 6));
        }
        if (o < 208) return sb[o - 192];
        return (uint8_t)(d >> (8 * (o - 208)));
    }
}

// ---- four output bytes at a time ----
// Every field of every block layout starts at a multiple of 4 bytes (only the trailing fp16 `d` of Q3_K / Q6_K is a
// 2-byte tail), and the strides the layouts combine (32, 64, 96, 128 input bytes) are multiples of 4 as well: output
// dword w of a block is a handful of LDS dword reads and byte-parallel mask / shift / or operations.
__device__ __forceinline__ uint32_t ld4(const uint8_t* p) { return *reinterpret_cast<const uint32_t*>(p); }
// four independent byte additions (no carry across bytes)
__device__ __forceinline__ uint32_t add4(uint32_t x, uint32_t c) {
    return ((x & 0x7f7f7f7fu) + (c & 0x7f7f7f7fu)) ^ ((x ^ c) & 0x80808080u);
}
template <int QT>
__device__ __forceinline__ uint32_t pack_dword(const uint8_t* q, const uint8_t* sb, const uint8_t* mb, uint16_t d,
                                               uint16_t dmin, int w) {
    const int o = 4 * w;
    auto hdr = [&](int j0) {  // four bytes of the 12-byte scale/min field of Q4_K / Q5_K
        return (uint32_t)pack_scale_min_byte(sb, mb, j0) | ((uint32_t)pack_scale_min_byte(sb, mb, j0 + 1) << 8) |
               ((uint32_t)pack_scale_min_byte(sb, mb, j0 + 2) << 16) | ((uint32_t)pack_scale_min_byte(sb, mb, j0 + 3) << 24);
    };
    if constexpr (QT == GQ_Q2_K) {  // scales[16] qs[64] d dmin
        if (o < 16) return (ld4(sb + o) & 0x0f0f0f0fu) | ((ld4(mb + o) & 0x0f0f0f0fu) << 4);
        if (o < 80) {
            const int t = o - 16, ch = t >> 5, l = t & 31;
            const uint8_t* c = q + ch * 128 + l;
            return ld4(c) | (ld4(c + 32) << 2) | (ld4(c + 64) << 4) | (ld4(c + 96) << 6);  // values 0..3: no masks
        }
        return (uint32_t)d | ((uint32_t)dmin << 16);
    } else if constexpr (QT == GQ_Q3_K) {  // hmask[32] qs[64] scales[12] | d
        if (o < 32) {
            uint32_t h = 0;
#pragma unroll
            for (int b = 0; b < 8; ++b) h |= ((add4(ld4(q + b * 32 + o), 0x04040404u) >> 2) & 0x01010101u) << b;  // (q+4) > 3
            return h;
        }
        if (o < 96) {
            const int t = o - 32, ch = t >> 5, l = t & 31;
            const uint8_t* c = q + ch * 128 + l;
            uint32_t r = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) r |= (add4(ld4(c + 32 * k), 0x04040404u) & 0x03030303u) << (2 * k);  // u > 3 ? u - 4 : u
            return r;
        }
        uint32_t r = 0;  // the 12 scale bytes: byte-wise (o = 96, 100, 104)
#pragma unroll
        for (int k = 0; k < 4; ++k) r |= (uint32_t)pack_byte<QT>(q, sb, mb, d, dmin, o + k) << (8 * k);
        return r;
    } else if constexpr (QT == GQ_Q4_K) {  // d dmin scales[12] qs[128]
        if (o == 0) return (uint32_t)d | ((uint32_t)dmin << 16);
        if (o < 16) return hdr(o - 4);
        const int t = o - 16, base = (t >> 5) * 64, l = t & 31;
        return ld4(q + base + l) | (ld4(q + base + 32 + l) << 4);  // values 0..15
    } else if constexpr (QT == GQ_Q5_K) {  // d dmin scales[12] qh[32] ql[128]
        if (o == 0) return (uint32_t)d | ((uint32_t)dmin << 16);
        if (o < 16) return hdr(o - 4);
        if (o < 48) {
            const int j = o - 16;
            uint32_t h = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                h |= ((ld4(q + 64 * k + j) >> 4) & 0x01010101u) << (2 * k);           // values 0..31: > 15 is bit 4
                h |= ((ld4(q + 64 * k + 32 + j) >> 4) & 0x01010101u) << (2 * k + 1);
            }
            return h;
        }
        const int t = o - 48, base = (t >> 5) * 64, j = t & 31;
        return (ld4(q + base + j) & 0x0f0f0f0fu) | ((ld4(q + base + 32 + j) & 0x0f0f0f0fu) << 4);
    } else {  // Q6_K  ql[128] qh[64] scales[16] | d ;  v = q + 32 in 0..63
        auto v4 = [&](int idx) { return add4(ld4(q + idx), 0x20202020u); };
        if (o < 128) {
            const int ch = o >> 6, t = o & 63, l = t & 31, hi = t >> 5, b = ch * 128 + l + 32 * hi;
            return (v4(b) & 0x0f0f0f0fu) | ((v4(b + 64) & 0x0f0f0f0fu) << 4);
        }
        if (o < 192) {
            const int t = o - 128, ch = t >> 5, l = t & 31, b = ch * 128 + l;
            return ((v4(b) >> 4) & 0x03030303u) | (((v4(b + 32) >> 4) & 0x03030303u) << 2) |
                   (((v4(b + 64) >> 4) & 0x03030303u) << 4) | (((v4(b + 96) >> 4) & 0x03030303u) << 6);
        }
        return ld4(sb + (o - 192));
    }
}

template <int QT, int TS, int NG>
__global__ __launch_bounds__(256) void pack_kernel(const uint8_t* __restrict__ qw, const uint16_t* __restrict__ d,
                                                   const uint8_t* __restrict__ s, const uint16_t* __restrict__ dmin,
                                                   const uint8_t* __restrict__ m, int64_t nblocks,
                                                   uint8_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint8_t sq[PB * 256];
    __shared__ __attribute__((aligned(16))) uint8_t so[PB * TS];
    __shared__ __attribute__((aligned(16))) uint8_t ss[PB * NG];
    __shared__ __attribute__((aligned(16))) uint8_t sm[PB * NG];
    __shared__ uint16_t sd[PB], sdm[PB];
    const int tid = threadIdx.x;
    for (int64_t b0 = (int64_t)blockIdx.x * PB; b0 < nblocks; b0 += (int64_t)gridDim.x * PB) {
        const int nb = (int)((nblocks - b0) < PB ? (nblocks - b0) : PB);
        // stage inputs: nb*256 bytes = nb*16 uint4 (two loads per thread before the first LDS write)
        {
            const uint4* src = reinterpret_cast<const uint4*>(qw + b0 * 256);
            const int n16 = nb * 16;
            uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0;
            if (tid < n16) v0 = src[tid];
            if (tid + 256 < n16) v1 = src[tid + 256];
            if (tid < n16) reinterpret_cast<uint4*>(sq)[tid] = v0;
            if (tid + 256 < n16) reinterpret_cast<uint4*>(sq)[tid + 256] = v1;
        }
        for (int t = tid; t < nb * NG; t += 256) {
            ss[t] = s[b0 * NG + t];
            sm[t] = m ? m[b0 * NG + t] : 0;
        }
        if (tid < nb) {
            sd[tid] = d[b0 + tid];
            sdm[tid] = dmin ? dmin[b0 + tid] : 0;
        }
        __syncthreads();
        constexpr int NDW = TS / 4;  // whole dwords of a block; TS % 4 == 2 (Q3_K, Q6_K): the fp16 d follows
        if constexpr (TS % 4 == 0) {
            // r05: 84- / 144- / 176-byte blocks are whole dwords and the workgroup's output region is contiguous: thread i's
            // dword IS output dword i -- stored straight to global memory (coalesced 4-byte stores), no LDS image of the
            // output, one barrier and one pass fewer
            uint32_t* op32 = reinterpret_cast<uint32_t*>(out + b0 * TS);
            if constexpr (QT == GQ_Q4_K || QT == GQ_Q5_K) {
                // the 4 header dwords of a block (d | dmin, 12 scale / min bytes assembled byte by byte) take a long path:
                // in one index space every wave carries a few of them and all its lanes wait -- body dwords first (one
                // uniform path), the headers in a pass of their own
                constexpr int HD = 4, BD = NDW - HD;
                for (int i = tid; i < nb * BD; i += 256) {
                    const int b = i / BD, w = HD + i % BD;
                    op32[b * NDW + w] = pack_dword<QT>(sq + b * 256, ss + b * NG, sm + b * NG, sd[b], sdm[b], w);
                }
                for (int i = tid; i < nb * HD; i += 256) {
                    const int b = i / HD, w = i % HD;
                    op32[b * NDW + w] = pack_dword<QT>(sq + b * 256, ss + b * NG, sm + b * NG, sd[b], sdm[b], w);
                }
            } else {
                for (int i = tid; i < nb * NDW; i += 256) {
                    const int b = i / NDW, w = i % NDW;
                    op32[i] = pack_dword<QT>(sq + b * 256, ss + b * NG, sm + b * NG, sd[b], sdm[b], w);
                }
            }
            __syncthreads();
            continue;
        }
        // (r05) one pass per KIND of dword -- Q3_K: hmask 8 | qs 16 | scales 3 (byte-wise); Q6_K: ql 32 | qh 16 | scales 4 -- so that
        // every wave runs one code path (in one index space each wave carried all kinds and its lanes waited for the longest)
        constexpr int W1 = QT == GQ_Q3_K ? 8 : (QT == GQ_Q6_K ? 32 : NDW), W2 = QT == GQ_Q3_K ? 24 : (QT == GQ_Q6_K ? 48 : NDW);
        auto pass = [&](int w0, int w1) {
            const int nw = w1 - w0;
            for (int i = tid; i < nb * nw; i += 256) {
                const int b = i / nw, w = w0 + i % nw;
                const uint32_t v = pack_dword<QT>(sq + b * 256, ss + b * NG, sm + b * NG, sd[b], sdm[b], w);
                uint8_t* dst = so + b * TS + 4 * w;
                if ((TS % 4 == 0) || !(b & 1)) {
                    *reinterpret_cast<uint32_t*>(dst) = v;
                } else {  // odd block of a 110- / 210-byte layout: 2-byte aligned
                    reinterpret_cast<uint16_t*>(dst)[0] = (uint16_t)v;
                    reinterpret_cast<uint16_t*>(dst)[1] = (uint16_t)(v >> 16);
                }
            }
        };
        pass(0, W1);
        if constexpr (W1 < NDW) pass(W1, W2);
        if constexpr (W2 < NDW) pass(W2, NDW);
        if constexpr (TS % 4 != 0) {
            if (tid < nb) *reinterpret_cast<uint16_t*>(so + tid * TS + 4 * NDW) = sd[tid];
        }
        __syncthreads();
        uint8_t* op = out + b0 * TS;  // b0 % 8 == 0 and 8*TS % 16 == 0 -> 16-byte aligned
        const int nbytes = nb * TS;
        for (int v = tid; v < nbytes / 16; v += 256)
            reinterpret_cast<uint4*>(op)[v] = reinterpret_cast<const uint4*>(so)[v];
        for (int t = (nbytes / 16) * 16 + tid; t < nbytes; t += 256) op[t] = so[t];
        __syncthreads();
    }
}

int launch_pack(int q_type, const uint8_t* q, const uint16_t* d, const uint8_t* s, const uint16_t* dmin,
                const uint8_t* m, int64_t R, int64_t C, uint8_t* out, hipStream_t st) {
    TypeInfo ti;
    if (!type_info(q_type, ti)) GQ_FAIL(GQ_E_BAD_TYPE, "gq_pack: unknown q_type %d", q_type);
    if (R <= 0 || C <= 0 || C % 256) GQ_FAIL(GQ_E_BAD_SHAPE, "gq_pack: R=%ld C=%ld (C %% 256 != 0)", (long)R, (long)C);
    if (!q || !d || !s || !out) GQ_FAIL(GQ_E_NULL, "gq_pack: null pointer");
    if (ti.k_search && (!dmin || !m)) GQ_FAIL(GQ_E_NULL, "gq_pack: dmin/m required for q_type %d", q_type);
    const int64_t nblocks = R * (C / 256);
    int64_t g = (nblocks + PB - 1) / PB;
    dim3 grid((unsigned)(g < 4096 ? g : 4096)), block(256);  // (1024 .. 16384 workgroups measured: no difference from 2048 up)
    ProfScope ps(PT_PACK, st);
    switch (q_type) {
    case GQ_Q2_K: hipLaunchKernelGGL((pack_kernel<GQ_Q2_K, 84, 16>), grid, block, 0, st, q, d, s, dmin, m, nblocks, out); break;
    case GQ_Q3_K: hipLaunchKernelGGL((pack_kernel<GQ_Q3_K, 110, 16>), grid, block, 0, st, q, d, s, dmin, m, nblocks, out); break;
    case GQ_Q4_K: hipLaunchKernelGGL((pack_kernel<GQ_Q4_K, 144, 8>), grid, block, 0, st, q, d, s, dmin, m, nblocks, out); break;
    case GQ_Q5_K: hipLaunchKernelGGL((pack_kernel<GQ_Q5_K, 176, 8>), grid, block, 0, st, q, d, s, dmin, m, nblocks, out); break;
    case GQ_Q6_K: hipLaunchKernelGGL((pack_kernel<GQ_Q6_K, 210, 16>), grid, block, 0, st, q, d, s, dmin, m, nblocks, out); break;
    }
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

}  // namespace gq
