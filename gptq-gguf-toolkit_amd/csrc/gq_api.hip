// gq_api.hip -- extern "C" surface of libgptqgguf_hip.so (see include/gptq_gguf.h).
#include "gq_common.hpp"

namespace gq {
thread_local char g_err[512] = "";

int launch_scale_search(const float*, int64_t, int64_t, int, const gq_search_t*, uint16_t*, int64_t, uint8_t*, int64_t,
                        uint16_t*, int64_t, uint8_t*, int64_t, hipStream_t, unsigned* panel = nullptr,
                        const int64_t* row_ends = nullptr, int nstack = 1);
int launch_dequantize(int, const uint8_t*, const uint16_t*, const uint8_t*, const uint16_t*, const uint8_t*, int64_t,
                      int64_t, void*, int, hipStream_t);
int launch_rtn_elementwise(const void*, int, const uint16_t*, const uint8_t*, const uint16_t*, const uint8_t*, int64_t,
                           int64_t, const TypeInfo&, uint8_t*, hipStream_t);
int launch_rtn_scale_search(const void*, int, int64_t, int64_t, int, const gq_search_t*, uint16_t*, uint8_t*, uint16_t*,
                            uint8_t*, hipStream_t);
int group_search(const void*, int, int64_t, int64_t, int, const gq_search_t*, float*, float*, uint16_t*, uint8_t*, uint16_t*,
                 uint8_t*, hipStream_t);
int launch_pack(int, const uint8_t*, const uint16_t*, const uint8_t*, const uint16_t*, const uint8_t*, int64_t, int64_t,
                uint8_t*, hipStream_t);
int launch_trailing_update(float*, int64_t, const float*, int64_t, const float*, int64_t, int64_t, int64_t, int64_t,
                           hipStream_t);
size_t gptq_workspace_bytes(int64_t, int64_t, int);
int gptq_quantize(float*, const float*, int64_t, int64_t, int, int, int, const gq_search_t*, uint8_t*, uint16_t*,
                  uint8_t*, uint16_t*, uint8_t*, void*, size_t, hipStream_t, const int32_t*, const int64_t* = nullptr, int = 1,
                  int32_t* = nullptr);
size_t h_accumulate_workspace_bytes(int64_t, int64_t);
int h_accumulate(float*, const void*, int, int64_t, int64_t, float, float, void*, size_t, hipStream_t);
int h_accumulate_grouped(int, float* const*, const void* const*, const int64_t*, const int64_t*, const float*, const float*,
                         int, void*, size_t, hipStream_t, const void* const* const*, const int64_t*);
size_t h_prepare_workspace_bytes(int64_t, int64_t);
int h_prepare(float*, float*, int64_t, int64_t, float, float*, int*, uint8_t*, void*, size_t, hipStream_t, bool);
int obq_quantize(float*, const float*, int64_t, int64_t, int, int, int, int, uint8_t*, float*, float*, void*, size_t,
                 hipStream_t);
int gptq_uses_helper_stream(int64_t, int64_t, int);
int far_helper_enable(int);
int w_prepare(const uint8_t*, float*, int64_t, int64_t, int*, hipStream_t);
int h_pack_upper(const float*, int64_t, float*, hipStream_t);
int h_unpack_upper(const float*, int64_t, float*, hipStream_t);
int h_stage(void*, const void*, int64_t, hipStream_t, int max_wgs = 0);
int h_stage_many(void*, const void* const*, const int64_t*, int, void*, size_t, hipStream_t);
size_t chol_gemm_workspace_bytes(int64_t, int64_t, int64_t);
int chol_gemm(float*, int64_t, const float*, int64_t, const float*, int64_t, int64_t, int64_t, int64_t, int, int, int, int, int,
              void*, size_t, hipStream_t);
int fwd_rmsnorm(const void*, const void*, void*, int64_t, int64_t, float, int, hipStream_t);
int fwd_rmsnorm_ordered(const void*, const void*, void*, int64_t, int64_t, float, int, float*, hipStream_t);
int fwd_rope(const void*, const void*, const void*, void*, int64_t, int, int, int, hipStream_t);
int fwd_silu_mul(const void*, const void*, void*, int64_t, int, hipStream_t);
}  // namespace gq

#include <algorithm>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include <atomic>
#include <cerrno>
#include <cstdlib>
#include <cstring>

namespace gq {
// ---- the option table (gq_common.hpp GQ_OPTION_LIST) ----
namespace {
struct OptEntry { const char* name; int64_t def, lo, hi; };
const OptEntry g_opt_table[OPT_COUNT] = {
#define GQ_X(name, def, lo, hi) {#name, (int64_t)(def), (int64_t)(lo), (int64_t)(hi)},
    GQ_OPTION_LIST(GQ_X)
#undef GQ_X
};
std::atomic<int64_t> g_opt[OPT_COUNT];
std::once_flag g_opt_once;
char g_opt_init_err[256];  // why GQ_OPTIONS did not parse ("" = fine); reported by every entry point through options_ok()
int opt_index(const char* name) {
    for (int i = 0; i < OPT_COUNT; ++i)
        if (!strcmp(g_opt_table[i].name, name)) return i;
    return -1;
}
// nullptr when `v` is acceptable for option i, else the reason
const char* opt_range_error(int i, int64_t v, char* buf, size_t n) {
    const OptEntry& e = g_opt_table[i];
    if (v < e.lo || v > e.hi) {
        snprintf(buf, n, "option '%s' = %lld is outside %lld..%lld", e.name, (long long)v, (long long)e.lo, (long long)e.hi);
        return buf;
    }
    if (i == OPT_la && (v & 1)) {
        snprintf(buf, n, "option 'la' = %lld must be even", (long long)v);
        return buf;
    }
    // a value the launcher would silently read as "no checkpoints" / an impossible super-tile is a typo, not a setting
    if (i == OPT_syrk_ck && v != 0 && ((v & (v - 1)) || v < 16)) {
        snprintf(buf, n, "option 'syrk_ck' = %lld must be 0 or a power of two >= 16", (long long)v);
        return buf;
    }
    if (i == OPT_syrk_gw && (v & (v - 1))) {
        snprintf(buf, n, "option 'syrk_gw' = %lld must be a power of two", (long long)v);
        return buf;
    }
    return nullptr;
}
void opt_init() {
    for (int i = 0; i < OPT_COUNT; ++i) g_opt[i].store(g_opt_table[i].def, std::memory_order_relaxed);
    const char* e = getenv("GQ_OPTIONS");  // "name=value,name=value": the library's ONE tuning variable
    if (!e) return;
    std::string s(e);
    size_t pos = 0;
    while (pos < s.size()) {
        size_t end = s.find(',', pos);
        if (end == std::string::npos) end = s.size();
        const std::string item = s.substr(pos, end - pos);
        pos = end + 1;
        const size_t eq = item.find('=');
        if (item.empty()) continue;
        const std::string key = eq == std::string::npos ? item : item.substr(0, eq);
        const int i = opt_index(key.c_str());
        // a typo must not silently measure the default: the first error is kept and EVERY entry point fails with it
        // (GQ_E_UNSUPPORTED + gq_last_error -> a Python exception with a traceback; no abort() inside a host process)
        if (i < 0) {
            if (!g_opt_init_err[0]) snprintf(g_opt_init_err, sizeof(g_opt_init_err), "GQ_OPTIONS names an unknown option '%s'", key.c_str());
            continue;
        }
        int64_t v = 1;
        if (eq != std::string::npos) {
            const char* txt = item.c_str() + eq + 1;
            char* endp = nullptr;
            errno = 0;
            v = strtoll(txt, &endp, 0);
            if (endp == txt || *endp != '\0' || errno == ERANGE) {
                if (!g_opt_init_err[0])
                    snprintf(g_opt_init_err, sizeof(g_opt_init_err), "GQ_OPTIONS: '%s' is not an integer value for option '%s'", txt, key.c_str());
                continue;
            }
        }
        char why[200];
        if (opt_range_error(i, v, why, sizeof(why))) {
            if (!g_opt_init_err[0]) snprintf(g_opt_init_err, sizeof(g_opt_init_err), "GQ_OPTIONS: %s", why);
            continue;
        }
        g_opt[i].store(v, std::memory_order_relaxed);
    }
}
}  // namespace
int64_t opt(Opt o) {
    std::call_once(g_opt_once, opt_init);
    return g_opt[o].load(std::memory_order_relaxed);
}
int options_ok() {
    std::call_once(g_opt_once, opt_init);
    if (g_opt_init_err[0]) GQ_FAIL(GQ_E_UNSUPPORTED, "%s", g_opt_init_err);
    return GQ_OK;
}

unsigned g_prof_mask = 0;
namespace {
struct Rec { hipEvent_t a, b; int tag; hipStream_t st; };
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
std::mutex g_mu;
hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace
void prof_begin(int tag, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_mu);
    Rec r{get_event(), get_event(), tag, st};
    (void)hipEventRecord(r.a, st);
    g_recs.push_back(r);
}
void prof_end(int tag, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (size_t i = g_recs.size(); i-- > 0;)
        if (g_recs[i].tag == tag) { (void)hipEventRecord(g_recs[i].b, st); break; }
}
}  // namespace gq

using namespace gq;

#define GQ_OPTIONS_OK()                               \
    do {                                              \
        if (int _rc = options_ok()) return _rc;       \
    } while (0)

extern "C" {

int gq_abi_version(void) { return GQ_ABI_VERSION; }

int gq_option_count(void) { return OPT_COUNT; }
const char* gq_option_name(int i) { return (i >= 0 && i < OPT_COUNT) ? g_opt_table[i].name : nullptr; }
int gq_option_get(const char* name, int64_t* value) {
    const int i = name ? opt_index(name) : -1;
    if (i < 0 || !value) GQ_FAIL(GQ_E_UNSUPPORTED, "gq_option_get: unknown option '%s'", name ? name : "(null)");
    GQ_OPTIONS_OK();
    *value = opt((Opt)i);
    return GQ_OK;
}
int gq_option_set(const char* name, int64_t value, int64_t* previous) {
    const int i = name ? opt_index(name) : -1;
    if (i < 0) GQ_FAIL(GQ_E_UNSUPPORTED, "gq_option_set: unknown option '%s'", name ? name : "(null)");
    const int64_t old = opt((Opt)i);  // (also runs the one-time initialisation)
    GQ_OPTIONS_OK();  // a GQ_OPTIONS that did not parse fails this entry point like every other
    char why[200];
    if (opt_range_error(i, value, why, sizeof(why))) GQ_FAIL(GQ_E_UNSUPPORTED, "gq_option_set: %s", why);
    g_opt[i].store(value, std::memory_order_relaxed);
    if (previous) *previous = old;
    return GQ_OK;
}
int gq_option_default(const char* name, int64_t* value) {
    const int i = name ? opt_index(name) : -1;
    if (i < 0 || !value) GQ_FAIL(GQ_E_UNSUPPORTED, "gq_option_default: unknown option '%s'", name ? name : "(null)");
    *value = g_opt_table[i].def;
    return GQ_OK;
}
const char* gq_last_error(void) { return g_err; }

int gq_type_info(int q_type, gq_type_info_t* out) {
    TypeInfo t;
    if (!out) GQ_FAIL(GQ_E_NULL, "gq_type_info: null out");
    if (!type_info(q_type, t)) GQ_FAIL(GQ_E_BAD_TYPE, "gq_type_info: unknown q_type %d", q_type);
    out->bits = t.bits; out->qmin = t.qmin; out->qmax = t.qmax; out->scale_maxq = t.scale_maxq;
    out->group = t.group; out->is_signed = t.is_signed; out->k_search = t.k_search; out->type_size = t.type_size;
    return GQ_OK;
}

size_t gq_workspace_bytes(int op, int64_t R, int64_t C, int64_t T, int block_size) {
    switch (op) {
    case GQ_WS_H_ACCUMULATE: return h_accumulate_workspace_bytes(T, C);
    case GQ_WS_H_PREPARE: return h_prepare_workspace_bytes(R, C);
    case GQ_WS_GPTQ_QUANTIZE: return gptq_workspace_bytes(R, C, block_size);
    case GQ_WS_CHOL_GEMM: return chol_gemm_workspace_bytes(R, C, T);
    default: return 0;
    }
}

int gq_h_accumulate(float* H, const void* X, int x_dtype, int64_t T, int64_t C, float beta, float alpha, void* ws,
                    size_t ws_bytes, void* stream) {
    GQ_OPTIONS_OK();
    return h_accumulate(H, X, x_dtype, T, C, beta, alpha, ws, ws_bytes, (hipStream_t)stream);
}

int gq_h_accumulate_grouped(int n, float* const* H_host, const void* const* X_host, const int64_t* T_host,
                            const int64_t* C_host, const float* beta_host, const float* alpha_host, int x_dtype,
                            void* ws, size_t ws_bytes, void* stream) {
    GQ_OPTIONS_OK();
    return h_accumulate_grouped(n, H_host, X_host, T_host, C_host, beta_host, alpha_host, x_dtype, ws, ws_bytes,
                                (hipStream_t)stream, nullptr, nullptr);
}

int gq_h_accumulate_segments(int n, float* const* H_host, const void* const* const* blocks_host, const int64_t* nblocks_host,
                             const int64_t* block_tokens_host, const int64_t* C_host, const float* beta_host,
                             const float* alpha_host, int x_dtype, void* ws, size_t ws_bytes, void* stream) {
    GQ_OPTIONS_OK();
    if (n <= 0 || n > 8) GQ_FAIL(GQ_E_BAD_SHAPE, "gq_h_accumulate_segments: n=%d not in 1..8", n);
    if (!blocks_host || !nblocks_host || !block_tokens_host) GQ_FAIL(GQ_E_NULL, "gq_h_accumulate_segments: null pointer");
    const void* X[8];
    int64_t T[8];
    for (int i = 0; i < n; ++i) {
        if (!blocks_host[i] || nblocks_host[i] <= 0 || block_tokens_host[i] <= 0)
            GQ_FAIL(GQ_E_BAD_SHAPE, "gq_h_accumulate_segments: problem %d has no blocks", i);
        X[i] = blocks_host[i][0];
        T[i] = nblocks_host[i] * block_tokens_host[i];
    }
    return h_accumulate_grouped(n, H_host, X, T, C_host, beta_host, alpha_host, x_dtype, ws, ws_bytes, (hipStream_t)stream,
                                blocks_host, nblocks_host);
}

int gq_h_prepare(float* H, float* W, int64_t R, int64_t C, float rel_damp, float* U, int* not_invertible,
                 uint8_t* col_flags_out, void* ws, size_t ws_bytes, void* stream) {
    GQ_OPTIONS_OK();
    return h_prepare(H, W, R, C, rel_damp, U, not_invertible, col_flags_out, ws, ws_bytes, (hipStream_t)stream, false);
}

int gq_gptq_uses_helper_stream(int64_t R, int64_t C, int block_size) { return gptq_uses_helper_stream(R, C, block_size); }
int gq_far_helper_enable(int on) { return far_helper_enable(on); }

int gq_obq_h_prepare(float* H, float* W, int64_t R, int64_t C, float rel_damp, float* U, int* not_invertible,
                     uint8_t* col_flags_out, void* ws, size_t ws_bytes, void* stream) {
    GQ_OPTIONS_OK();
    return h_prepare(H, W, R, C, rel_damp, U, not_invertible, col_flags_out, ws, ws_bytes, (hipStream_t)stream, true);
}

int gq_obq_quantize(float* W, const float* U, int64_t R, int64_t C, int bits, int group_size, int sym, int block_size,
                    uint8_t* qweight, float* scale, float* zero, void* ws, size_t ws_bytes, void* stream) {
    GQ_OPTIONS_OK();
    return obq_quantize(W, U, R, C, bits, group_size, sym, block_size, qweight, scale, zero, ws, ws_bytes,
                        (hipStream_t)stream);
}

int gq_w_prepare(const uint8_t* col_flags, float* W, int64_t R, int64_t C, int* mismatch, void* stream) {
    GQ_OPTIONS_OK();
    return w_prepare(col_flags, W, R, C, mismatch, (hipStream_t)stream);
}

int gq_h_stage(void* dst, const void* src, int64_t nbytes, void* stream) {
    GQ_OPTIONS_OK();
    return h_stage(dst, src, nbytes, (hipStream_t)stream);
}
int gq_h_stage_many(void* dst, const void* const* srcs_host, const int64_t* nbytes_host, int n, void* ws, size_t ws_bytes,
                    void* stream) {
    GQ_OPTIONS_OK();
    return h_stage_many(dst, srcs_host, nbytes_host, n, ws, ws_bytes, (hipStream_t)stream);
}
int gq_h_pack_upper(const float* H, int64_t C, float* buf, void* stream) {
    GQ_OPTIONS_OK();
    return h_pack_upper(H, C, buf, (hipStream_t)stream);
}
int gq_h_unpack_upper(const float* buf, int64_t C, float* H, void* stream) {
    GQ_OPTIONS_OK();
    return h_unpack_upper(buf, C, H, (hipStream_t)stream);
}

int gq_scale_search(const float* x, int64_t rows, int64_t ld, int q_type, const gq_search_t* p, uint16_t* d,
                    int64_t d_stride, uint8_t* s, int64_t s_ld, uint16_t* dmin, int64_t dmin_stride, uint8_t* m,
                    int64_t m_ld, void* stream) {
    GQ_OPTIONS_OK();
    if (!x || !d || !s || !dmin || !m) GQ_FAIL(GQ_E_NULL, "gq_scale_search: null pointer");
    return launch_scale_search(x, rows, ld, q_type, p, d, d_stride, s, s_ld, dmin, dmin_stride, m, m_ld,
                               (hipStream_t)stream);
}

int gq_group_search(const void* x, int x_dtype, int64_t rows, int64_t ld, int q_type, const gq_search_t* p,
                    float* group_scale, float* group_zero, uint16_t* d, uint8_t* s, uint16_t* dmin, uint8_t* m,
                    void* stream) {
    GQ_OPTIONS_OK();
    return group_search(x, x_dtype, rows, ld, q_type, p, group_scale, group_zero, d, s, dmin, m, (hipStream_t)stream);
}

int gq_gptq_quantize(float* W, const float* U, int64_t R, int64_t C, int q_type, int block_size, int static_groups,
                     const gq_search_t* p, uint8_t* qweight, uint16_t* d, uint8_t* s, uint16_t* dmin, uint8_t* m,
                     void* ws, size_t ws_bytes, void* stream) {
    GQ_OPTIONS_OK();
    return gptq_quantize(W, U, R, C, q_type, block_size, static_groups, p, qweight, d, s, dmin, m, ws, ws_bytes,
                         (hipStream_t)stream, nullptr);
}

int gq_gptq_quantize_slice(float* W, const float* U, int64_t R, int64_t C, int q_type, int block_size, int static_groups,
                           const gq_search_t* p, uint8_t* qweight, uint16_t* d, uint8_t* s, uint16_t* dmin, uint8_t* m,
                           int32_t* panel_researches, void* ws, size_t ws_bytes, void* stream) {
    GQ_OPTIONS_OK();
    if (!panel_researches) GQ_FAIL(GQ_E_NULL, "gq_gptq_quantize_slice: null panel_researches");
    return gptq_quantize(W, U, R, C, q_type, block_size, static_groups, p, qweight, d, s, dmin, m, ws, ws_bytes,
                         (hipStream_t)stream, nullptr, nullptr, 1, panel_researches);
}

int gq_gptq_quantize_stacked(float* W, const float* U, int64_t R, int64_t C, int q_type, int block_size, int static_groups,
                             const gq_search_t* p, uint8_t* qweight, uint16_t* d, uint8_t* s, uint16_t* dmin, uint8_t* m,
                             const int64_t* row_ends_host, int n_stacked, void* ws, size_t ws_bytes, void* stream) {
    GQ_OPTIONS_OK();
    if (n_stacked < 1 || (n_stacked > 1 && !row_ends_host)) GQ_FAIL(GQ_E_NULL, "gq_gptq_quantize_stacked: n_stacked=%d without row_ends", n_stacked);
    return gptq_quantize(W, U, R, C, q_type, block_size, static_groups, p, qweight, d, s, dmin, m, ws, ws_bytes,
                         (hipStream_t)stream, nullptr, row_ends_host, n_stacked);
}

int gq_gptq_quantize_perm(float* W, const float* U, int64_t R, int64_t C, int q_type, int block_size, const int32_t* perm,
                          const uint16_t* d, const uint8_t* s, const uint16_t* dmin, const uint8_t* m, uint8_t* qweight,
                          void* ws, size_t ws_bytes, void* stream) {
    GQ_OPTIONS_OK();
    if (!perm) GQ_FAIL(GQ_E_NULL, "gq_gptq_quantize_perm: null perm");
    return gptq_quantize(W, U, R, C, q_type, block_size, 1, nullptr, qweight, const_cast<uint16_t*>(d), const_cast<uint8_t*>(s),
                         const_cast<uint16_t*>(dmin), const_cast<uint8_t*>(m), ws, ws_bytes, (hipStream_t)stream, perm);
}

int gq_rtn_quantize(const void* W, int w_dtype, int64_t R, int64_t C, int q_type, const gq_search_t* p,
                    uint8_t* qweight, uint16_t* d, uint8_t* s, uint16_t* dmin, uint8_t* m, void* stream) {
    GQ_OPTIONS_OK();
    TypeInfo ti;
    if (!type_info(q_type, ti)) GQ_FAIL(GQ_E_BAD_TYPE, "gq_rtn_quantize: unknown q_type %d", q_type);
    if (R <= 0 || C <= 0 || C % 256) GQ_FAIL(GQ_E_BAD_SHAPE, "gq_rtn_quantize: R=%ld C=%ld", (long)R, (long)C);
    if (!W || !qweight || !d || !s || !dmin || !m) GQ_FAIL(GQ_E_NULL, "gq_rtn_quantize: null pointer");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (w_dtype == GQ_F32) {
        const float* Wf = (const float*)W;
        const int64_t ng = C / ti.group, nsg = C / 256;
        const int gps = 256 / ti.group;
        for (int64_t c = 0; c < C; c += 256)  // quantizer.py:300-310
            if ((rc = launch_scale_search(Wf + c, R, C, q_type, p, d + c / 256, nsg, s + (c / 256) * gps, ng,
                                          dmin + c / 256, nsg, m + (c / 256) * gps, ng, st)))
                return rc;
    } else {
        // the reference runs make_*quants in the model dtype (quantizer.py:109,195)
        if ((rc = launch_rtn_scale_search(W, w_dtype, R, C, q_type, p, d, s, dmin, m, st))) return rc;
    }
    return launch_rtn_elementwise(W, w_dtype, d, s, dmin, m, R, C, ti, qweight, st);
}

int gq_dequantize(int q_type, const uint8_t* qweight, const uint16_t* d, const uint8_t* s, const uint16_t* dmin,
                  const uint8_t* m, int64_t R, int64_t C, void* out, int out_dtype, void* stream) {
    GQ_OPTIONS_OK();
    if (!qweight || !d || !s || !dmin || !m || !out) GQ_FAIL(GQ_E_NULL, "gq_dequantize: null pointer");
    return launch_dequantize(q_type, qweight, d, s, dmin, m, R, C, out, out_dtype, (hipStream_t)stream);
}

int gq_pack(int q_type, const uint8_t* qweight, const uint16_t* d, const uint8_t* s, const uint16_t* dmin,
            const uint8_t* m, int64_t R, int64_t C, uint8_t* out, void* stream) {
    GQ_OPTIONS_OK();
    return launch_pack(q_type, qweight, d, s, dmin, m, R, C, out, (hipStream_t)stream);
}

int gq_trailing_update(float* Cmat, int64_t ldc, const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M,
                       int64_t N, int64_t K, void* stream) {
    GQ_OPTIONS_OK();
    if (!Cmat || !A || !B) GQ_FAIL(GQ_E_NULL, "gq_trailing_update: null pointer");
    return launch_trailing_update(Cmat, ldc, A, lda, B, ldb, M, N, K, (hipStream_t)stream);
}

int gq_chol_gemm(float* Cmat, int64_t ldc, const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M, int64_t N,
                 int64_t K, int trans_b, int mode, int k_range, int lower, int planes, void* ws, size_t ws_bytes, void* stream) {
    GQ_OPTIONS_OK();
    return chol_gemm(Cmat, ldc, A, lda, B, ldb, M, N, K, trans_b, mode, k_range, lower, planes, ws, ws_bytes,
                     (hipStream_t)stream);
}

int gq_stage_to_host(void* host_dst, const void* src, int64_t nbytes, void* stream) {
    GQ_OPTIONS_OK();
    if (nbytes == 0) return GQ_OK;
    if (!host_dst || !src) GQ_FAIL(GQ_E_NULL, "gq_stage_to_host: null pointer");
    void* dptr = nullptr;
    GQ_HIP(hipHostGetDevicePointer(&dptr, host_dst, 0));  // fails unless host_dst is pinned (hipHostRegister / hipHostMalloc)
    return h_stage(dptr, src, nbytes, (hipStream_t)stream, (int)opt(OPT_stage_host_wgs));
}

int gq_fwd_rmsnorm(const void* x, const void* weight, void* out, int64_t tokens, int64_t C, float eps, int dtype, void* stream) {
    GQ_OPTIONS_OK();
    return fwd_rmsnorm(x, weight, out, tokens, C, eps, dtype, (hipStream_t)stream);
}
int gq_fwd_rmsnorm_ordered(const void* x, const void* weight, void* out, int64_t tokens, int64_t C, float eps, int dtype,
                           float* stats, void* stream) {
    GQ_OPTIONS_OK();
    return fwd_rmsnorm_ordered(x, weight, out, tokens, C, eps, dtype, stats, (hipStream_t)stream);
}
int gq_fwd_rope(const void* x, const void* cos_, const void* sin_, void* out, int64_t tokens, int heads, int head_dim, int dtype,
                void* stream) {
    GQ_OPTIONS_OK();
    return fwd_rope(x, cos_, sin_, out, tokens, heads, head_dim, dtype, (hipStream_t)stream);
}
int gq_fwd_silu_mul(const void* gate, const void* up, void* out, int64_t n, int dtype, void* stream) {
    GQ_OPTIONS_OK();
    return fwd_silu_mul(gate, up, out, n, dtype, (hipStream_t)stream);
}

// ---- profiling (bench.py): HIP-event timing of selected kernels on their launch stream ----
void gq_prof_enable(unsigned tag_mask) { g_prof_mask = tag_mask; }
int gq_prof_ntags(void) { return PT_COUNT; }
const char* gq_prof_name(int tag) {
    static const char* names[PT_COUNT] = {"transpose16", "syrk", "prepare_elementwise", "diag_potrf_inv", "potrf_gemm32",
                                          "trtri_gemm32", "scale_search", "gptq_segment", "trailing_gemm32",
                                          "block_far_update", "dequantize", "rtn_quantize", "pack", "trailing_far_gemm32",
                                          "chol_image_gemm", "chol_image_split"};
    return (tag >= 0 && tag < PT_COUNT) ? names[tag] : "?";
}
/* synchronises the recorded events; ms[tag], n[tag] accumulate; records are recycled.
 * busy_ms[tag] (optional) accumulates the length of the UNION of the tag's launch intervals: launches of
 * one tag that overlap on different streams are counted once (ms[tag] counts the overlap twice). */
int gq_prof_collect2(double* ms_host, long* n_host, double* busy_ms_host) {
    std::lock_guard<std::mutex> lk(g_mu);
    std::vector<std::pair<double, double>> iv[PT_COUNT];
    // GQ_PROF_DUMP=<file>: append every interval as "tag name stream start_ms end_ms" (timeline studies)
    const char* dump_path = getenv("GQ_PROF_DUMP");
    FILE* dump = (dump_path && busy_ms_host) ? fopen(dump_path, "a") : nullptr;
    for (auto& r : g_recs) {
        float t = 0.f, t0 = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
            ms_host[r.tag] += t;
            n_host[r.tag] += 1;
            if (busy_ms_host && hipEventElapsedTime(&t0, g_recs.front().a, r.a) == hipSuccess) {
                iv[r.tag].emplace_back((double)t0, (double)t0 + t);
                if (dump) fprintf(dump, "%d %s %p %.4f %.4f\n", r.tag, gq_prof_name(r.tag), (void*)r.st, t0, t0 + t);
            }
        }
    }
    if (dump) {
        fprintf(dump, "# end of collect\n");
        fclose(dump);
    }
    if (busy_ms_host)
        for (int tag = 0; tag < PT_COUNT; ++tag) {
            std::sort(iv[tag].begin(), iv[tag].end());
            double end = -1e300;
            for (auto& x : iv[tag]) {
                if (x.second <= end) continue;
                busy_ms_host[tag] += x.second - (x.first > end ? x.first : end);
                end = x.second;
            }
        }
    for (auto& r : g_recs) {
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_recs.clear();
    return GQ_OK;
}
int gq_prof_collect(double* ms_host, long* n_host) { return gq_prof_collect2(ms_host, n_host, nullptr); }

}  // extern "C"
