// gq_cholsub.hpp -- K3, the bottom of the recursion as ONE resident launch per sub-problem (r06).
// Included by gq_cholesky.hip only (after gq_gemm32.hpp and gq_diag5.hpp).
//
// Below the image-GEMM levels gq_h_prepare used to be a string of ~53 dependent launches per 1792-wide diagonal
// sub-problem (14 leaves of 32 us on ONE workgroup + 39 fp32 GEMMs of a few 64x64 tiles): 424 launches for C = 14336.
// Alone that costs their sum; inside a block's step every one of them queues for a CU behind the other chains' far-GEMM
// tiles (r05 timeline: 12.9 ms for the small GEMMs against 3.3 alone).  A resident kernel cannot be queued.
//
// The sub-problem (chol_inv_rec's subtree below a node of <= chol_sub blocks whose products are all gemm32 ones) becomes
// a static task graph executed by G resident workgroups:
//   * phases = exactly the launches the recursion would make (leaf; G1: L21 = A21 X11^T; G2: A22 -= L21 L21^T; G3:
//     A21 <- L21 X11; G4: X21 = -X22 A21), with the SAME tile functions (gemm32_tile<.., 64, FULL>, diag_blk5_body): every
//     output element is the same k-ordered chain as before, so U does not change by a bit (test_h_prepare_* unchanged);
//   * a phase depends on the earlier phases whose blocks it reads or overwrites (host: region analysis on the 128-block
//     grid, transitively reduced -- at most two per phase) instead of on "everything before it": leaves run next to
//     the inverse's products of the previous node;
//   * tasks (a phase's tiles in chunks) sit in one list in the recursion's order; a workgroup CLAIMS the next one with
//     an atomic, waits for the phase's dependencies (done counters), runs it, releases (device-scope: the 8 XCDs do not
//     share an L2) and counts it done.  Claiming makes the kernel safe with ANY number of resident workgroups: a task
//     is only ever waited for after it has been claimed by a running workgroup, and the list is a topological order.
#pragma once

#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace gq {
namespace csub {

enum : uint32_t { T_LEAF = 0, T_G1 = 1, T_G2 = 2, T_G3 = 3, T_G4 = 4 };
constexpr int PH_WORDS = 8;     // per phase: type | C coords | A coords | B coords | gy, gx | K | ntasks, tiles/task | deps
constexpr int HDR_WORDS = 4;    // nphases, ntasks, -, -
constexpr int CNT_STRIDE = 128; // counter words per sub-problem launch: [0] next task, [1] who gave up, [2] give-up flag, [32 + p] tasks of phase p done
constexpr int MAX_BLOCKS = 16;  // widest sub-problem (phases <= 16 + 3 * 15 = 61 < 96 counters)
constexpr size_t LDS_BYTES = DIAG5_LDS + 16;  // the leaf's image + the claimed task's slot behind it
constexpr unsigned POLL_LIMIT = 1u << 21;     // polls of one wait before the executor gives up (a few seconds)

struct HostPhase {
    uint32_t type;
    int cr, cc, ar, ac, br, bc;  // block coordinates (relative to the sub-problem) of C, A, B
    int gy, gx, K;               // 64-tiles, K in elements
    int ntiles;
    struct Rect { int m, r0, r1, c0, c1; };
    std::vector<Rect> reads, writes;
    std::vector<int> deps;
};

inline bool rect_hit(const HostPhase::Rect& a, const HostPhase::Rect& b) {
    return a.m == b.m && a.r0 < b.r1 && b.r0 < a.r1 && a.c0 < b.c1 && b.c0 < a.c1;
}
inline bool conflicts(const HostPhase& q, const HostPhase& p) {  // q earlier than p
    for (const auto& w : q.writes) {
        for (const auto& r : p.reads) if (rect_hit(w, r)) return true;
        for (const auto& r : p.writes) if (rect_hit(w, r)) return true;
    }
    for (const auto& r : q.reads)
        for (const auto& w : p.writes) if (rect_hit(r, w)) return true;
    return false;
}

// the recursion of chol_inv_rec on blocks [lo, hi) of the sub-problem (same split: mid = (lo + hi) / 2)
inline void collect(std::vector<HostPhase>& ph, int lo, int hi) {
    enum { MA = 0, MX = 1, MT = 2 };
    if (hi - lo == 1) {
        HostPhase p{};
        p.type = T_LEAF; p.cr = p.cc = lo; p.gy = p.gx = 1; p.K = NB; p.ntiles = 1;
        p.reads = {{MA, lo, hi, lo, hi}};
        p.writes = {{MX, lo, hi, lo, hi}};
        ph.push_back(p);
        return;
    }
    const int mid = (lo + hi) / 2, n1 = mid - lo, n2 = hi - mid;
    collect(ph, lo, mid);
    {   // G1: Tmp21 = A21 X11^T            gemm32 <true, 1, false, 1>
        HostPhase p{};
        p.type = T_G1; p.cr = mid; p.cc = lo; p.ar = mid; p.ac = lo; p.br = lo; p.bc = lo;
        p.gy = 2 * n2; p.gx = 2 * n1; p.K = n1 * NB; p.ntiles = p.gy * p.gx;
        p.writes = {{MT, mid, hi, lo, mid}};
        p.reads = {{MA, mid, hi, lo, mid}, {MX, lo, mid, lo, mid}};
        ph.push_back(p);
    }
    {   // G2: A22 -= Tmp21 Tmp21^T (lower) gemm32 <true, 0, true, 0>
        HostPhase p{};
        p.type = T_G2; p.cr = mid; p.cc = mid; p.ar = mid; p.ac = lo; p.br = mid; p.bc = lo;
        p.gy = 2 * n2; p.gx = 2 * n2; p.K = n1 * NB; p.ntiles = p.gx * (p.gx + 1) / 2;
        p.writes = {{MA, mid, hi, mid, hi}};
        p.reads = {{MT, mid, hi, lo, mid}, {MA, mid, hi, mid, hi}};
        ph.push_back(p);
    }
    {   // G3: A21 <- Tmp21 X11            gemm32 <false, 1, false, 2>
        HostPhase p{};
        p.type = T_G3; p.cr = mid; p.cc = lo; p.ar = mid; p.ac = lo; p.br = lo; p.bc = lo;
        p.gy = 2 * n2; p.gx = 2 * n1; p.K = n1 * NB; p.ntiles = p.gy * p.gx;
        p.writes = {{MA, mid, hi, lo, mid}};
        p.reads = {{MT, mid, hi, lo, mid}, {MX, lo, mid, lo, mid}};
        ph.push_back(p);
    }
    collect(ph, mid, hi);
    {   // G4: X21 = -X22 A21              gemm32 <false, 2, false, 3>
        HostPhase p{};
        p.type = T_G4; p.cr = mid; p.cc = lo; p.ar = mid; p.ac = mid; p.br = mid; p.bc = lo;
        p.gy = 2 * n2; p.gx = 2 * n1; p.K = n2 * NB; p.ntiles = p.gy * p.gx;
        p.writes = {{MX, mid, hi, lo, mid}};
        p.reads = {{MX, mid, hi, mid, hi}, {MA, mid, hi, lo, mid}};
        ph.push_back(p);
    }
}

// device image of the plan of a sub-problem of nb blocks executed by up to `wgs` workgroups
inline std::vector<uint32_t> make_plan(int nb, int wgs) {
    std::vector<HostPhase> ph;
    collect(ph, 0, nb);
    const int n = (int)ph.size();
    std::vector<std::vector<char>> clo(n, std::vector<char>(n, 0));  // clo[p][q]: p depends (transitively) on q
    for (int p = 0; p < n; ++p) {
        std::vector<int> d;
        for (int q = 0; q < p; ++q)
            if (conflicts(ph[q], ph[p])) d.push_back(q);
        for (int q : d) {
            clo[p][q] = 1;
            for (int r = 0; r < q; ++r) if (clo[q][r]) clo[p][r] = 1;
        }
        for (int q : d) {  // keep q unless another direct dependency already implies it
            bool implied = false;
            for (int q2 : d) if (q2 != q && clo[q2][q]) implied = true;
            if (!implied) ph[p].deps.push_back(q);
        }
        if (ph[p].deps.size() > 4) fprintf(stderr, "gq: sub-problem phase with %zu dependencies\n", ph[p].deps.size()), abort();
    }
    std::vector<uint32_t> w(HDR_WORDS + (size_t)PH_WORDS * n, 0u);
    std::vector<uint32_t> tasks;
    for (int p = 0; p < n; ++p) {
        HostPhase& h = ph[p];
        // 64x64 tiles put four times as many workgroups on a small product; a phase with more 64-tiles than workgroups
        // takes 128x128 tiles instead (a quarter of the tiles, each with twice the operand reuse: measured on the 896 | 896
        // node of a 1792-wide sub-problem, 48 workgroups: 215 us of 64-tiles).  Same k-ordered chain per element either way.
        uint32_t ts128 = 0;
        if (h.type != T_LEAF && h.ntiles > wgs) {
            ts128 = 1;
            h.gy /= 2; h.gx /= 2;
            h.ntiles = h.type == T_G2 ? h.gx * (h.gx + 1) / 2 : h.gy * h.gx;
        }
        // chunks: about two tasks per workgroup and phase at most -- a task pays one claim and one device-scope release
        int tpt = (h.ntiles + 2 * wgs - 1) / (2 * wgs);
        if (tpt < 1) tpt = 1;
        const int ntasks = (h.ntiles + tpt - 1) / tpt;
        uint32_t* d = w.data() + HDR_WORDS + (size_t)PH_WORDS * p;
        d[0] = h.type | (ts128 << 4);
        d[1] = (uint32_t)h.cr | ((uint32_t)h.cc << 8);
        d[2] = (uint32_t)h.ar | ((uint32_t)h.ac << 8);
        d[3] = (uint32_t)h.br | ((uint32_t)h.bc << 8);
        d[4] = (uint32_t)h.gy | ((uint32_t)h.gx << 16);
        d[5] = (uint32_t)h.K;
        d[6] = (uint32_t)ntasks | ((uint32_t)tpt << 16);
        uint32_t dw = 0xffffffffu;
        for (size_t i = 0; i < h.deps.size(); ++i) dw = (dw & ~(0xffu << (8 * i))) | ((uint32_t)h.deps[i] << (8 * i));
        d[7] = dw;
        for (int c = 0; c < ntasks; ++c) tasks.push_back(((uint32_t)p << 16) | (uint32_t)c);
    }
    w[0] = (uint32_t)n;
    w[1] = (uint32_t)tasks.size();
    w.insert(w.end(), tasks.begin(), tasks.end());
    while (w.size() % 64) w.push_back(0u);
    return w;
}

// ---- the executor ----
__device__ __forceinline__ uint32_t ld_cnt(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// tiles [t0, t1) of one product, TS x TS each, by the calling workgroup (waves 0-3 compute, wave 4 keeps the barriers)
template <int TS>
__device__ __forceinline__ void run_tiles(uint32_t kind, float* A, float* X, float* T, int64_t ld, uint32_t wc, uint32_t wa,
                                          uint32_t wb, unsigned gy, unsigned gx, uint32_t K, unsigned t0, unsigned t1) {
    auto at = [&](float* M, uint32_t w) { return M + (int64_t)(w & 0xffu) * NB * ld + (int64_t)(w >> 8) * NB; };
    const int64_t M = (int64_t)gy * TS, N = (int64_t)gx * TS;
    if (kind == T_G1) {
        float* Cb = at(T, wc); const float* Ab = at(A, wa); const float* Bb = at(X, wb);
        for (unsigned id = t0; id < t1; ++id)
            gemm32_tile<true, 1, false, 1, 0, TS, true, 4, true>(Cb, ld, Ab, ld, Bb, ld, M, N, K, id % gx, id / gx, gx, gy);
    } else if (kind == T_G2) {
        float* Cb = at(A, wc); const float* Ab = at(T, wa);
        for (unsigned id = t0; id < t1; ++id) {
            // lower tile (by, bx), bx <= by, from the linear index id = by (by + 1) / 2 + bx
            unsigned by = (unsigned)((__fsqrt_rn(8.0f * (float)id + 1.0f) - 1.0f) * 0.5f);
            while ((by + 1) * (by + 2) / 2 <= id) ++by;
            while (by * (by + 1) / 2 > id) --by;
            const unsigned bx = id - by * (by + 1) / 2;
            gemm32_tile<true, 0, true, 0, 0, TS, true, 4, true>(Cb, ld, Ab, ld, Ab, ld, N, N, K, bx, by, gx, gx);
        }
    } else if (kind == T_G3) {
        float* Cb = at(A, wc); const float* Ab = at(T, wa); const float* Bb = at(X, wb);
        for (unsigned id = t0; id < t1; ++id)
            gemm32_tile<false, 1, false, 2, 0, TS, true, 4, true>(Cb, ld, Ab, ld, Bb, ld, M, N, K, id % gx, id / gx, gx, gy);
    } else {
        float* Cb = at(X, wc); const float* Ab = at(X, wa); const float* Bb = at(A, wb);
        for (unsigned id = t0; id < t1; ++id)
            gemm32_tile<false, 2, false, 3, 0, TS, true, 4, true>(Cb, ld, Ab, ld, Bb, ld, M, N, K, id % gx, id / gx, gx, gy);
    }
}

// (inlined on purpose: as a non-inlined callee -- which would keep the leaf's own register allocation, 200 VGPRs / 97 SGPRs
// without a spill -- the kernel hangs on this stack.)
__device__ __forceinline__ void leaf_task(float* A, float* Xo, int64_t ld, int* flag) {
    diag_blk5_body(A, ld, Xo, ld, flag);
}

__global__ __launch_bounds__(320) void chol_sub_kernel(float* __restrict__ A, float* __restrict__ X, float* __restrict__ T,
                                                       int64_t ld, int* __restrict__ flag, const uint32_t* __restrict__ plan,
                                                       uint32_t* __restrict__ cnt) {
    extern __shared__ __attribute__((aligned(16))) float csub_smem[];
    volatile uint32_t* slot = reinterpret_cast<volatile uint32_t*>(reinterpret_cast<unsigned char*>(csub_smem) + DIAG5_LDS);
    const int tid = threadIdx.x;
    const uint32_t nph = plan[0], ntask = plan[1];
    const uint32_t* phw = plan + HDR_WORDS;
    const uint32_t* tasks = phw + (size_t)PH_WORDS * nph;
    uint32_t* done = cnt + 32;
    auto claim = [&]() {  // lane 0 only
        uint32_t c = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ld_cnt(cnt + 2)) c = 0xffffffffu;  // another workgroup gave up: leave
        slot[0] = c;
        slot[1] = 0u;
    };
    // Every lane-0-only region is bracketed by barriers and every loop exit is on a wave-uniform SGPR value: with the claim at
    // the loop's head and the release at its tail the compiler merged the two lane-0 regions across the back edge and let lane 0
    // leave the (to it, divergent) loop while lanes 1-63 of its wave ran into the next iteration's barrier -- a hang.
    if (tid == 0) claim();
    __syncthreads();
    for (;;) {
        const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot[0]);
        // [probe: claimed]
        if (t >= ntask) break;
        const uint32_t tw = tasks[t], p = tw >> 16, chunk = tw & 0xffffu;
        const uint32_t* d = phw + (size_t)PH_WORDS * p;
        const uint32_t type = d[0], wc = d[1], wa = d[2], wb = d[3], gyx = d[4], K = d[5], nt = d[6], deps = d[7];
        if (tid == 0) {
            // A dependency is only ever waited for after all its tasks were claimed by running workgroups, so the wait is
            // bounded by their run time.  The poll limit (seconds) is a fuse against a fault elsewhere hanging the GPU:
            // it raises the not-invertible flag (U = I, reported by the caller) and makes every workgroup leave.
            unsigned polls = 0;
            for (int i = 0; i < 4 && !slot[1]; ++i) {
                const uint32_t q = (deps >> (8 * i)) & 0xffu;
                if (q == 0xffu) break;
                const uint32_t need = phw[(size_t)PH_WORDS * q + 6] & 0xffffu;
                while (ld_cnt(done + q) < need) {
                    if (++polls > POLL_LIMIT || ld_cnt(cnt + 2)) {
                        __hip_atomic_store(cnt + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(cnt + 1, 0xdead0000u | (p << 8) | q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        *flag = 1;
                        slot[1] = 1u;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(4);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        // [probe: dependencies met]
        if (__builtin_amdgcn_readfirstlane((int)slot[1])) break;
        const unsigned gy = gyx & 0xffffu, gx = gyx >> 16;
        auto at = [&](float* M, uint32_t w) { return M + (int64_t)(w & 0xffu) * NB * ld + (int64_t)(w >> 8) * NB; };
        const uint32_t kind = type & 0xfu;
        if (kind == T_LEAF) {
            leaf_task(at(A, wc), at(X, wc), ld, flag);
        } else {
            const unsigned tpt = nt >> 16;
            const unsigned ntiles = (kind == T_G2) ? gx * (gx + 1) / 2 : gx * gy;
            unsigned t0 = chunk * tpt, t1 = t0 + tpt;
            if (t1 > ntiles) t1 = ntiles;
            if (type & 0x10u) run_tiles<128>(kind, A, X, T, ld, wc, wa, wb, gy, gx, K, t0, t1);
            else run_tiles<64>(kind, A, X, T, ld, wc, wa, wb, gy, gx, K, t0, t1);
        }
        // every wave's stores have reached the L2 (the barrier waits for vmcnt 0); ONE device-scope release writes the
        // L2 back for the other XCDs, then the task counts as done
        __syncthreads();
        // [probe: work done]
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(done + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            claim();
        }
        // [probe: released]
        __syncthreads();
    }
}

struct SubPlans {
    std::vector<uint32_t> words;               // plans of the distinct sub-problem sizes, then the zeroed counters
    std::map<int, size_t> plan_at;             // nb -> word offset of its plan
    size_t cnt_at = 0;                         // word offset of the first counter block
    int nsub = 0;
};

}  // namespace csub
}  // namespace gq
