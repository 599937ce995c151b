// gq_forward.hip -- SURVEY 8(f) row 2, the calibration forward path: the elementwise third of a Llama decoder layer as
// three HBM-bound kernels (opt-in: Quantizer(fused_forward=True) / --fused_forward; the GEMMs and the attention stay
// with the framework).  HF's eager modules (transformers models/llama/modeling_llama.py) spell each of these as 2-8
// torch elementwise kernels over [tokens, hidden] tensors with fp32 round trips; per block forward of 2048 tokens that
// is 0.36 ms of 1.16 ms.  Each kernel reproduces the module's arithmetic operation by operation -- fp32
// compute, round to the tensor dtype after every torch op -- so the outputs equal HF eager's except for the summation
// order of the RMSNorm mean:
//   rmsnorm   LlamaRMSNorm.forward:  h = float(x); var = mean(h^2); h = h * rsqrt(var + eps); out = w * dtype(h)
//   rope      apply_rotary_pos_emb:  q' = dtype(q cos) + dtype(rotate_half(q) sin), the same for k
//   silu_mul  LlamaMLP.forward:      dtype(silu(gate)) * up
// One pass each, 16-byte accesses.  Algorithmic bytes per token: rmsnorm 2 * 2C, rope 2 * 2 (Hq + Hkv) D (+ cos/sin),
// silu_mul 3 * 2 * I.
#include "gq_common.hpp"

namespace gq {

template <bool BF16> __device__ __forceinline__ float ld16(uint16_t v) { return BF16 ? bf2f(v) : h2f(v); }
template <bool BF16> __device__ __forceinline__ uint16_t st16(float f) { return BF16 ? f2bf(f) : f2h(f); }

// one workgroup (256 threads) per row; C % 8 == 0
template <bool BF16>
__global__ __launch_bounds__(256) void fwd_rmsnorm_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                          uint16_t* __restrict__ out, int64_t C, float eps) {
    const int64_t row = blockIdx.x;
    const uint16_t* xr = x + row * C;
    __shared__ float part[4];
    float acc = 0.0f;
    for (int64_t c = 8 * (int64_t)threadIdx.x; c < C; c += 8 * 256) {
        const uint4 v = *reinterpret_cast<const uint4*>(xr + c);
        const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = ld16<BF16>((uint16_t)(u[e] & 0xffff)), b = ld16<BF16>((uint16_t)(u[e] >> 16));
            acc = fmaf(a, a, acc);
            acc = fmaf(b, b, acc);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    const float var = ((part[0] + part[1]) + (part[2] + part[3])) / (float)C;
    const float r = rsqrtf(var + eps);  // torch.rsqrt (fp32): the same device function ATen calls
    for (int64_t c = 8 * (int64_t)threadIdx.x; c < C; c += 8 * 256) {
        const uint4 v = *reinterpret_cast<const uint4*>(xr + c), wv = *reinterpret_cast<const uint4*>(w + c);
        const uint32_t u[4] = {v.x, v.y, v.z, v.w}, ww[4] = {wv.x, wv.y, wv.z, wv.w};
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // hidden_states * rsqrt(...) in fp32, .to(input_dtype), then weight * that (rounded to the dtype again)
            const float h0 = ld16<BF16>(st16<BF16>(ld16<BF16>((uint16_t)(u[e] & 0xffff)) * r));
            const float h1 = ld16<BF16>(st16<BF16>(ld16<BF16>((uint16_t)(u[e] >> 16)) * r));
            const uint16_t o0 = st16<BF16>(ld16<BF16>((uint16_t)(ww[e] & 0xffff)) * h0);
            const uint16_t o1 = st16<BF16>(ld16<BF16>((uint16_t)(ww[e] >> 16)) * h1);
            o[e] = (uint32_t)o0 | ((uint32_t)o1 << 16);
        }
        *reinterpret_cast<uint4*>(out + row * C + c) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// Correctly rounded 1 / sqrt(x) in fp32 (what torch.rsqrt returns on this stack: 0 differences in 4 M samples against the
// double-precision value rounded once; v_rsq_f32 alone is a 1-ulp approximation).  The double-precision estimate rounded to
// fp32 is right unless 1 / sqrt(x) lies within 2^-53 of a rounding boundary; the boundaries y -+ ulp / 2 are tested exactly:
// (y +- h)^2 has <= 50 significant bits (exact in double), its product with x is taken as an unevaluated sum t + e (fma).
__device__ __forceinline__ float rsqrt_rn(float x) {
    float y = (float)(1.0 / sqrt((double)x));
    if (!(x > 0.0f) || !(y > 0.0f) || y > 3.0e38f) return y;  // zero / negative / nan / overflow: nothing to repair
    // product (m * m) * x compared with 1, exactly
    auto above_one = [](double m, double xd) {  // m^2 x > 1 ?
        const double p = m * m;                 // exact
        const double t = p * xd, e = fma(p, xd, -t);
        return t > 1.0 || (t == 1.0 && e > 0.0);
    };
    auto below_one = [](double m, double xd) {  // m^2 x < 1 ?
        const double p = m * m;
        const double t = p * xd, e = fma(p, xd, -t);
        return t < 1.0 || (t == 1.0 && e < 0.0);
    };
    const float up = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, y) + 1u);
    const float dn = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, y) - 1u);
    const double hi = 0.5 * ((double)y + (double)up), lo = 0.5 * ((double)y + (double)dn);  // the rounding boundaries
    // 1 / sqrt(x) > hi  <=>  hi^2 x < 1: the neighbour above is closer; 1 / sqrt(x) < lo  <=>  lo^2 x > 1: the one below
    // (on a boundary exactly -- impossible for 1 / sqrt of a float other than a power of four -- y stays)
    if (below_one(hi, (double)x)) return up;
    if (above_one(lo, (double)x)) return dn;
    return y;
}

// The same module with the statistics summed in the order ATen's reduce kernel uses for mean(-1) of a contiguous fp32
// [rows >= 8, C] tensor on a 64-wide wavefront (aten/src/ATen/native/cuda/Reduce.cuh; confirmed by emulating the order
// with torch ops, profiles/aten_mean_order_probe.py):
//   C / 64 < 128 (C <= 5120 ...): 64 lanes per row; lane x adds the squares of elements (x + 64 it) 4 + j, it = 0, 1, ...,
//     into FOUR accumulators j = 0..3 (vectorised loads), folds them as ((a0 + a1) + a2) + a3, then the lanes are folded
//     with shuffle-down offsets 1, 2, 4, ..., 32;
//   C / 64 >= 128 (C >= 8192): the row is also split over 8 wavefronts: thread (x, y) takes elements
//     (x + 64 y + 512 it) 4 + j; every wavefront folds its lanes as above, then the eight wavefront sums are combined with
//     offsets 4, 2, 1 (y += y + offset) -- found by searching the order space (profiles/aten_mean_order_probe2.py).
// The mean is the sum times fl(rows / (rows C)).  With that order and rsqrt_rn the output is bit-identical to HF eager's
// -- which the Python side VERIFIES on the first call of every (C, dtype) before it trusts this kernel
// (forward_fused.py); a PyTorch that reduces in another order simply keeps the eager module.  C % 512 == 0.
template <bool BF16>
__device__ __forceinline__ void rmsnorm_apply_row(const uint16_t* __restrict__ xr, const uint16_t* __restrict__ w,
                                                  uint16_t* __restrict__ outr, int64_t C, float r, int t0, int nt) {
    for (int64_t c = 8 * (int64_t)t0; c < C; c += 8 * (int64_t)nt) {
        const uint4 v = *reinterpret_cast<const uint4*>(xr + c), wv = *reinterpret_cast<const uint4*>(w + c);
        const uint32_t u[4] = {v.x, v.y, v.z, v.w}, ww[4] = {wv.x, wv.y, wv.z, wv.w};
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float h0 = ld16<BF16>(st16<BF16>(ld16<BF16>((uint16_t)(u[e] & 0xffff)) * r));
            const float h1 = ld16<BF16>(st16<BF16>(ld16<BF16>((uint16_t)(u[e] >> 16)) * r));
            const uint16_t o0 = st16<BF16>(ld16<BF16>((uint16_t)(ww[e] & 0xffff)) * h0);
            const uint16_t o1 = st16<BF16>(ld16<BF16>((uint16_t)(ww[e] >> 16)) * h1);
            o[e] = (uint32_t)o0 | ((uint32_t)o1 << 16);
        }
        *reinterpret_cast<uint4*>(outr + c) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}
template <bool BF16>
__device__ __forceinline__ float rmsnorm_thread_sum(const uint16_t* __restrict__ xr, int64_t C, int t0, int nt) {
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    for (int64_t c = 4 * (int64_t)t0; c < C; c += 4 * (int64_t)nt) {
        const uint2 v = *reinterpret_cast<const uint2*>(xr + c);
        const float f0 = ld16<BF16>((uint16_t)(v.x & 0xffff)), f1 = ld16<BF16>((uint16_t)(v.x >> 16));
        const float f2 = ld16<BF16>((uint16_t)(v.y & 0xffff)), f3 = ld16<BF16>((uint16_t)(v.y >> 16));
        a0 = a0 + f0 * f0;  // pow(2) rounds the square to fp32, the reduction adds it: two roundings (-ffp-contract=off)
        a1 = a1 + f1 * f1;
        a2 = a2 + f2 * f2;
        a3 = a3 + f3 * f3;
    }
    return ((a0 + a1) + a2) + a3;
}
// The non-split form with the row kept in registers between the two passes (r06; C = 256 IT, IT <= 20): lane x holds its
// IT 4-element pieces -- the very pieces ATen's order gives it for the sum -- loaded with IT independent 8-byte loads, and
// applies weight and 1 / rms to THEM (the apply is elementwise: which lane scales an element does not change a bit).  One
// read of x instead of two and all loads in flight at once: 16.6 -> ~9 us per 2048 x 4096 call, 16 128 calls per 8B model.
template <bool BF16, int IT>
__global__ __launch_bounds__(256) void fwd_rmsnorm_ordered_reg_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                                      uint16_t* __restrict__ out, int64_t rows, float eps, float factor,
                                                                      float* __restrict__ stats) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wid;
    if (row >= rows) return;
    constexpr int64_t C = 256 * IT;
    const uint16_t* xr = x + row * C;
    uint2 v[IT], wv[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) v[it] = *reinterpret_cast<const uint2*>(xr + 4 * (lane + 64 * it));
#pragma unroll
    for (int it = 0; it < IT; ++it) wv[it] = *reinterpret_cast<const uint2*>(w + 4 * (lane + 64 * it));
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll
    for (int it = 0; it < IT; ++it) {  // rmsnorm_thread_sum's order
        const float f0 = ld16<BF16>((uint16_t)(v[it].x & 0xffff)), f1 = ld16<BF16>((uint16_t)(v[it].x >> 16));
        const float f2 = ld16<BF16>((uint16_t)(v[it].y & 0xffff)), f3 = ld16<BF16>((uint16_t)(v[it].y >> 16));
        a0 = a0 + f0 * f0;
        a1 = a1 + f1 * f1;
        a2 = a2 + f2 * f2;
        a3 = a3 + f3 * f3;
    }
    float s = ((a0 + a1) + a2) + a3;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) s = s + __shfl_down(s, o);
    const float var = __shfl(s, 0) * factor;
    const float r = rsqrt_rn(var + eps);
    if (stats && lane == 0) {
        stats[2 * row] = var;
        stats[2 * row + 1] = r;
    }
    uint16_t* outr = out + row * C;
#pragma unroll
    for (int it = 0; it < IT; ++it) {  // rmsnorm_apply_row's arithmetic per element
        const uint32_t u[2] = {v[it].x, v[it].y}, ww[2] = {wv[it].x, wv[it].y};
        uint32_t o[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float h0 = ld16<BF16>(st16<BF16>(ld16<BF16>((uint16_t)(u[e] & 0xffff)) * r));
            const float h1 = ld16<BF16>(st16<BF16>(ld16<BF16>((uint16_t)(u[e] >> 16)) * r));
            const uint16_t o0 = st16<BF16>(ld16<BF16>((uint16_t)(ww[e] & 0xffff)) * h0);
            const uint16_t o1 = st16<BF16>(ld16<BF16>((uint16_t)(ww[e] >> 16)) * h1);
            o[e] = (uint32_t)o0 | ((uint32_t)o1 << 16);
        }
        *reinterpret_cast<uint2*>(outr + 4 * (lane + 64 * it)) = make_uint2(o[0], o[1]);
    }
}

// SPLIT = false: one wavefront per row, four rows per workgroup of 256;  true: one row per workgroup of 512
template <bool BF16, bool SPLIT>
__global__ __launch_bounds__(SPLIT ? 512 : 256) void fwd_rmsnorm_ordered_kernel(const uint16_t* __restrict__ x,
                                                                                const uint16_t* __restrict__ w,
                                                                                uint16_t* __restrict__ out, int64_t rows,
                                                                                int64_t C, float eps, float factor,
                                                                                float* __restrict__ stats) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __shared__ float part[8];
    __shared__ float rr;
    if constexpr (!SPLIT) {
        const int64_t row = (int64_t)blockIdx.x * 4 + wid;
        if (row >= rows) return;
        float s = rmsnorm_thread_sum<BF16>(x + row * C, C, lane, 64);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) s = s + __shfl_down(s, o);
        const float var = __shfl(s, 0) * factor;
        const float r = rsqrt_rn(var + eps);
        if (stats && lane == 0) {
            stats[2 * row] = var;
            stats[2 * row + 1] = r;
        }
        rmsnorm_apply_row<BF16>(x + row * C, w, out + row * C, C, r, lane, 64);
    } else {
        const int64_t row = blockIdx.x;
        float s = rmsnorm_thread_sum<BF16>(x + row * C, C, (int)threadIdx.x, 512);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) s = s + __shfl_down(s, o);
        if (lane == 0) part[wid] = s;
        __syncthreads();
        if (threadIdx.x == 0) {  // the eight wavefront sums: offsets 4, 2, 1
            const float v0 = part[0] + part[4], v1 = part[1] + part[5], v2 = part[2] + part[6], v3 = part[3] + part[7];
            const float u0 = v0 + v2, u1 = v1 + v3;
            const float var = (u0 + u1) * factor;
            rr = rsqrt_rn(var + eps);
            if (stats) {
                stats[2 * row] = var;
                stats[2 * row + 1] = rr;
            }
        }
        __syncthreads();
        rmsnorm_apply_row<BF16>(x + row * C, w, out + row * C, C, rr, (int)threadIdx.x, 512);
    }
}

// x: [tokens, heads, D] (the memory layout of q_proj(x).view(B, L, H, D)); cos / sin: [tokens, D]; D % 16 == 0.
// thread = 8 consecutive d of the first half of one (token, head) and their partners in the second half
template <bool BF16>
__global__ __launch_bounds__(256) void fwd_rope_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ cs,
                                                       const uint16_t* __restrict__ sn, uint16_t* __restrict__ out,
                                                       int64_t tokens, int heads, int D) {
    const int per = D / 16;  // threads per (token, head)
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= tokens * heads * per) return;
    const int64_t th = t / per, tok = th / heads;
    const int d0 = (int)(t % per) * 8, half = D / 2;
    const uint16_t* xp = x + th * D;
    const uint4 a4 = *reinterpret_cast<const uint4*>(xp + d0), b4 = *reinterpret_cast<const uint4*>(xp + half + d0);
    const uint4 ca = *reinterpret_cast<const uint4*>(cs + tok * D + d0), cb = *reinterpret_cast<const uint4*>(cs + tok * D + half + d0);
    const uint4 sa = *reinterpret_cast<const uint4*>(sn + tok * D + d0), sb = *reinterpret_cast<const uint4*>(sn + tok * D + half + d0);
    const uint32_t A[4] = {a4.x, a4.y, a4.z, a4.w}, B[4] = {b4.x, b4.y, b4.z, b4.w};
    const uint32_t CA[4] = {ca.x, ca.y, ca.z, ca.w}, CB[4] = {cb.x, cb.y, cb.z, cb.w};
    const uint32_t SA[4] = {sa.x, sa.y, sa.z, sa.w}, SB[4] = {sb.x, sb.y, sb.z, sb.w};
    uint32_t oa[4], ob[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        uint16_t ra[2], rb[2];
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
            const int sh = 16 * hlf;
            const float a = ld16<BF16>((uint16_t)(A[e] >> sh)), b = ld16<BF16>((uint16_t)(B[e] >> sh));
            // first half d:  q[d] cos[d] + (-q[d + D/2]) sin[d];  second half: q[d + D/2] cos[d + D/2] + q[d] sin[d + D/2]
            const float p0 = ld16<BF16>(st16<BF16>(a * ld16<BF16>((uint16_t)(CA[e] >> sh))));
            const float p1 = ld16<BF16>(st16<BF16>(-b * ld16<BF16>((uint16_t)(SA[e] >> sh))));
            const float p2 = ld16<BF16>(st16<BF16>(b * ld16<BF16>((uint16_t)(CB[e] >> sh))));
            const float p3 = ld16<BF16>(st16<BF16>(a * ld16<BF16>((uint16_t)(SB[e] >> sh))));
            ra[hlf] = st16<BF16>(p0 + p1);
            rb[hlf] = st16<BF16>(p2 + p3);
        }
        oa[e] = (uint32_t)ra[0] | ((uint32_t)ra[1] << 16);
        ob[e] = (uint32_t)rb[0] | ((uint32_t)rb[1] << 16);
    }
    uint16_t* op = out + th * D;
    *reinterpret_cast<uint4*>(op + d0) = make_uint4(oa[0], oa[1], oa[2], oa[3]);
    *reinterpret_cast<uint4*>(op + half + d0) = make_uint4(ob[0], ob[1], ob[2], ob[3]);
}

// out = dtype(silu(gate)) * up, n % 8 == 0
template <bool BF16>
__global__ __launch_bounds__(256) void fwd_silu_mul_kernel(const uint16_t* __restrict__ gate, const uint16_t* __restrict__ up,
                                                           uint16_t* __restrict__ out, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const uint4 g4 = *reinterpret_cast<const uint4*>(gate + 8 * i), u4 = *reinterpret_cast<const uint4*>(up + 8 * i);
        const uint32_t G[4] = {g4.x, g4.y, g4.z, g4.w}, U[4] = {u4.x, u4.y, u4.z, u4.w};
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint16_t r[2];
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
                const float g = ld16<BF16>((uint16_t)(G[e] >> (16 * hlf))), u = ld16<BF16>((uint16_t)(U[e] >> (16 * hlf)));
                const float s = ld16<BF16>(st16<BF16>(g / (1.0f + expf(-g))));  // ATen silu: x / (1 + exp(-x)) in fp32
                r[hlf] = st16<BF16>(s * u);
            }
            o[e] = (uint32_t)r[0] | ((uint32_t)r[1] << 16);
        }
        *reinterpret_cast<uint4*>(out + 8 * i) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

static bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

int fwd_rmsnorm(const void* x, const void* w, void* out, int64_t T, int64_t C, float eps, int dtype, hipStream_t st) {
    if (!x || !w || !out) GQ_FAIL(GQ_E_NULL, "gq_fwd_rmsnorm: null pointer");
    if (T <= 0 || C <= 0 || C % 8 || !al16(x) || !al16(w) || !al16(out)) GQ_FAIL(GQ_E_BAD_SHAPE, "gq_fwd_rmsnorm: T=%ld C=%ld (C %% 8, 16-byte alignment)", (long)T, (long)C);
    if (dtype != GQ_F16 && dtype != GQ_BF16) GQ_FAIL(GQ_E_BAD_TYPE, "gq_fwd_rmsnorm: dtype %d (fp16 / bf16)", dtype);
    if (dtype == GQ_BF16) hipLaunchKernelGGL(fwd_rmsnorm_kernel<true>, dim3((unsigned)T), dim3(256), 0, st, (const uint16_t*)x, (const uint16_t*)w, (uint16_t*)out, C, eps);
    else hipLaunchKernelGGL(fwd_rmsnorm_kernel<false>, dim3((unsigned)T), dim3(256), 0, st, (const uint16_t*)x, (const uint16_t*)w, (uint16_t*)out, C, eps);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}
int fwd_rmsnorm_ordered(const void* x, const void* w, void* out, int64_t T, int64_t C, float eps, int dtype, float* stats,
                        hipStream_t st) {
    if (!x || !w || !out) GQ_FAIL(GQ_E_NULL, "gq_fwd_rmsnorm_ordered: null pointer");
    if (T <= 0 || C <= 0 || C % 512 || !al16(x) || !al16(w) || !al16(out))
        GQ_FAIL(GQ_E_BAD_SHAPE, "gq_fwd_rmsnorm_ordered: T=%ld C=%ld (C %% 512, 16-byte alignment)", (long)T, (long)C);
    if (dtype != GQ_F16 && dtype != GQ_BF16) GQ_FAIL(GQ_E_BAD_TYPE, "gq_fwd_rmsnorm_ordered: dtype %d (fp16 / bf16)", dtype);
    const float factor = (float)T / (float)(T * C);  // ATen's mean: float(outputs) / inputs
    const bool split = C / 64 >= 128;                 // Reduce.cuh: values per thread >= block height (8) x 16
    const dim3 grid((unsigned)(split ? T : (T + 3) / 4)), block(split ? 512 : 256);
    if (!split && (C == 4096 || C == 2048 || C == 5120)) {  // the widths of the Llama family below 8192: the row stays in registers
#define GQ_RN_REG(B16, IT_) \
    hipLaunchKernelGGL((fwd_rmsnorm_ordered_reg_kernel<B16, IT_>), grid, block, 0, st, (const uint16_t*)x, (const uint16_t*)w, (uint16_t*)out, T, eps, factor, stats)
        const bool b16 = dtype == GQ_BF16;
        if (C == 4096) { if (b16) GQ_RN_REG(true, 16); else GQ_RN_REG(false, 16); }
        else if (C == 2048) { if (b16) GQ_RN_REG(true, 8); else GQ_RN_REG(false, 8); }
        else { if (b16) GQ_RN_REG(true, 20); else GQ_RN_REG(false, 20); }
#undef GQ_RN_REG
        GQ_LAUNCH_CHECK();
        return GQ_OK;
    }
#define GQ_RN_LAUNCH(B16, SP) \
    hipLaunchKernelGGL((fwd_rmsnorm_ordered_kernel<B16, SP>), grid, block, 0, st, (const uint16_t*)x, (const uint16_t*)w, (uint16_t*)out, T, C, eps, factor, stats)
    if (dtype == GQ_BF16) { if (split) GQ_RN_LAUNCH(true, true); else GQ_RN_LAUNCH(true, false); }
    else { if (split) GQ_RN_LAUNCH(false, true); else GQ_RN_LAUNCH(false, false); }
#undef GQ_RN_LAUNCH
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}
int fwd_rope(const void* x, const void* cos_, const void* sin_, void* out, int64_t tokens, int heads, int D, int dtype, hipStream_t st) {
    if (!x || !cos_ || !sin_ || !out) GQ_FAIL(GQ_E_NULL, "gq_fwd_rope: null pointer");
    if (tokens <= 0 || heads <= 0 || D <= 0 || D % 16 || !al16(x) || !al16(cos_) || !al16(sin_) || !al16(out))
        GQ_FAIL(GQ_E_BAD_SHAPE, "gq_fwd_rope: tokens=%ld heads=%d D=%d (D %% 16, 16-byte alignment)", (long)tokens, heads, D);
    if (dtype != GQ_F16 && dtype != GQ_BF16) GQ_FAIL(GQ_E_BAD_TYPE, "gq_fwd_rope: dtype %d (fp16 / bf16)", dtype);
    const int64_t n = tokens * heads * (D / 16);
    const dim3 grid((unsigned)((n + 255) / 256));
    if (dtype == GQ_BF16) hipLaunchKernelGGL(fwd_rope_kernel<true>, grid, dim3(256), 0, st, (const uint16_t*)x, (const uint16_t*)cos_, (const uint16_t*)sin_, (uint16_t*)out, tokens, heads, D);
    else hipLaunchKernelGGL(fwd_rope_kernel<false>, grid, dim3(256), 0, st, (const uint16_t*)x, (const uint16_t*)cos_, (const uint16_t*)sin_, (uint16_t*)out, tokens, heads, D);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}
int fwd_silu_mul(const void* gate, const void* up, void* out, int64_t n, int dtype, hipStream_t st) {
    if (!gate || !up || !out) GQ_FAIL(GQ_E_NULL, "gq_fwd_silu_mul: null pointer");
    if (n <= 0 || n % 8 || !al16(gate) || !al16(up) || !al16(out)) GQ_FAIL(GQ_E_BAD_SHAPE, "gq_fwd_silu_mul: n=%ld (n %% 8, 16-byte alignment)", (long)n);
    if (dtype != GQ_F16 && dtype != GQ_BF16) GQ_FAIL(GQ_E_BAD_TYPE, "gq_fwd_silu_mul: dtype %d (fp16 / bf16)", dtype);
    const int64_t n8 = n / 8;
    const dim3 grid((unsigned)((n8 + 255) / 256 < 16384 ? (n8 + 255) / 256 : 16384));
    if (dtype == GQ_BF16) hipLaunchKernelGGL(fwd_silu_mul_kernel<true>, grid, dim3(256), 0, st, (const uint16_t*)gate, (const uint16_t*)up, (uint16_t*)out, n8);
    else hipLaunchKernelGGL(fwd_silu_mul_kernel<false>, grid, dim3(256), 0, st, (const uint16_t*)gate, (const uint16_t*)up, (uint16_t*)out, n8);
    GQ_LAUNCH_CHECK();
    return GQ_OK;
}

}  // namespace gq
