"""ctypes binding of libgptqgguf_hip.so (include/gptq_gguf.h).

This is the ONLY compute backend of the package: there is no CPU or torch fallback.
If the shared library is missing or a call fails, the error is raised to the caller.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.environ.get("GQ_SO_PATH") or os.path.join(CSRC, "libgptqgguf_hip.so")  # override: kernel A/B probes

ABI_VERSION = 6  # include/gptq_gguf.h GQ_ABI_VERSION this binding was written against
F32, F16, BF16 = 0, 1, 2
WS_H_ACCUMULATE, WS_H_PREPARE, WS_GPTQ_QUANTIZE, WS_CHOL_GEMM = 1, 2, 3, 4

EXPORTS = (
    "gq_abi_version", "gq_last_error", "gq_option_count", "gq_option_name", "gq_option_get", "gq_option_default", "gq_option_set", "gq_type_info", "gq_workspace_bytes", "gq_h_accumulate", "gq_h_accumulate_grouped", "gq_h_accumulate_segments", "gq_h_stage", "gq_h_stage_many", "gq_h_prepare", "gq_w_prepare", "gq_h_pack_upper", "gq_h_unpack_upper",
    "gq_scale_search", "gq_group_search", "gq_gptq_quantize", "gq_gptq_quantize_stacked", "gq_gptq_quantize_slice", "gq_gptq_quantize_perm", "gq_gptq_uses_helper_stream", "gq_far_helper_enable", "gq_obq_h_prepare", "gq_obq_quantize", "gq_rtn_quantize", "gq_dequantize", "gq_pack", "gq_trailing_update", "gq_chol_gemm", "gq_stage_to_host", "gq_fwd_rmsnorm", "gq_fwd_rmsnorm_ordered", "gq_fwd_rope", "gq_fwd_silu_mul",
    "gq_prof_enable", "gq_prof_ntags", "gq_prof_name", "gq_prof_collect", "gq_prof_collect2",
)


class GQError(RuntimeError):
    pass


class TypeInfo(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in
                ("bits", "qmin", "qmax", "scale_maxq", "group", "is_signed", "k_search", "type_size")]


class Search(ctypes.Structure):
    _fields_ = [("rmin", ctypes.c_double), ("rdelta", ctypes.c_double), ("nstep", ctypes.c_int),
                ("quant_scale", ctypes.c_int), ("grid", ctypes.c_int), ("maxshrink", ctypes.c_double)]


def build(force: bool = False) -> str:
    """Compile the HIP sources for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-s", "-j8"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return SO_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise GQError(
            f"{SO_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (or `make -C {CSRC}`). There is no CPU fallback.")
    # torch first: its wheel bundles its own libamdhip64, and the library must bind to THAT runtime (the one
    # that owns the device context and the streams of the tensors it is handed).  Loaded before torch, the
    # library pulls /opt/rocm's copy in and every call fails with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    L = ctypes.CDLL(SO_PATH)
    vp, i64, ci, cf, sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
    sp = ctypes.POINTER(Search)
    L.gq_abi_version.restype = ci
    if L.gq_abi_version() != ABI_VERSION:  # a stale build (or GQ_SO_PATH) must fail here, not at a symbol lookup later
        raise GQError(f"{SO_PATH} has ABI version {L.gq_abi_version()}, this package binds version {ABI_VERSION}: rebuild "
                      f"(`make -C {CSRC}`)")
    L.gq_last_error.restype = ctypes.c_char_p
    L.gq_option_name.restype = ctypes.c_char_p
    L.gq_option_name.argtypes = [ci]
    L.gq_option_get.argtypes = [ctypes.c_char_p, ctypes.POINTER(i64)]
    L.gq_option_default.argtypes = [ctypes.c_char_p, ctypes.POINTER(i64)]
    L.gq_option_set.argtypes = [ctypes.c_char_p, i64, ctypes.POINTER(i64)]
    L.gq_type_info.argtypes = [ci, ctypes.POINTER(TypeInfo)]
    L.gq_workspace_bytes.argtypes = [ci, i64, i64, i64, ci]
    L.gq_workspace_bytes.restype = sz
    L.gq_h_accumulate.argtypes = [vp, vp, ci, i64, i64, cf, cf, vp, sz, vp]
    L.gq_h_accumulate_grouped.argtypes = [ci, vp, vp, vp, vp, vp, vp, ci, vp, sz, vp]
    L.gq_h_accumulate_segments.argtypes = [ci, vp, vp, vp, vp, vp, vp, vp, ci, vp, sz, vp]
    L.gq_h_stage.argtypes = [vp, vp, i64, vp]
    L.gq_h_stage_many.argtypes = [vp, vp, vp, ci, vp, sz, vp]
    L.gq_h_prepare.argtypes = [vp, vp, i64, i64, cf, vp, vp, vp, vp, sz, vp]
    L.gq_w_prepare.argtypes = [vp, vp, i64, i64, vp, vp]
    L.gq_h_pack_upper.argtypes = [vp, i64, vp, vp]
    L.gq_h_unpack_upper.argtypes = [vp, i64, vp, vp]
    L.gq_scale_search.argtypes = [vp, i64, i64, ci, sp, vp, i64, vp, i64, vp, i64, vp, i64, vp]
    L.gq_group_search.argtypes = [vp, ci, i64, i64, ci, sp, vp, vp, vp, vp, vp, vp, vp]
    L.gq_gptq_quantize.argtypes = [vp, vp, i64, i64, ci, ci, ci, sp, vp, vp, vp, vp, vp, vp, sz, vp]
    L.gq_gptq_quantize_stacked.argtypes = [vp, vp, i64, i64, ci, ci, ci, sp, vp, vp, vp, vp, vp, vp, ci, vp, sz, vp]
    L.gq_gptq_quantize_slice.argtypes = [vp, vp, i64, i64, ci, ci, ci, sp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    L.gq_gptq_quantize_perm.argtypes = [vp, vp, i64, i64, ci, ci, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    L.gq_gptq_uses_helper_stream.argtypes = [i64, i64, ci]
    L.gq_far_helper_enable.argtypes = [ci]
    L.gq_obq_h_prepare.argtypes = L.gq_h_prepare.argtypes
    L.gq_obq_quantize.argtypes = [vp, vp, i64, i64, ci, ci, ci, ci, vp, vp, vp, vp, sz, vp]
    L.gq_rtn_quantize.argtypes = [vp, ci, i64, i64, ci, sp, vp, vp, vp, vp, vp, vp]
    L.gq_dequantize.argtypes = [ci, vp, vp, vp, vp, vp, i64, i64, vp, ci, vp]
    L.gq_pack.argtypes = [ci, vp, vp, vp, vp, vp, i64, i64, vp, vp]
    L.gq_trailing_update.argtypes = [vp, i64, vp, i64, vp, i64, i64, i64, i64, vp]
    L.gq_chol_gemm.argtypes = [vp, i64, vp, i64, vp, i64, i64, i64, i64, ci, ci, ci, ci, ci, vp, sz, vp]
    L.gq_stage_to_host.argtypes = [vp, vp, i64, vp]
    L.gq_fwd_rmsnorm.argtypes = [vp, vp, vp, i64, i64, cf, ci, vp]
    L.gq_fwd_rmsnorm_ordered.argtypes = [vp, vp, vp, i64, i64, cf, ci, vp, vp]
    L.gq_fwd_rope.argtypes = [vp, vp, vp, vp, i64, ci, ci, ci, vp]
    L.gq_fwd_silu_mul.argtypes = [vp, vp, vp, i64, ci, vp]
    L.gq_prof_enable.argtypes = [ctypes.c_uint]
    L.gq_prof_enable.restype = None
    L.gq_prof_name.argtypes = [ci]
    L.gq_prof_name.restype = ctypes.c_char_p
    L.gq_prof_collect.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long)]
    L.gq_prof_collect2.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long),
                                   ctypes.POINTER(ctypes.c_double)]
    for name in EXPORTS:
        getattr(L, name)  # raises AttributeError if the .so lacks a declared symbol
    _lib = L
    return L


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().gq_last_error().decode(errors="replace")
        raise GQError(f"{what} failed (status {rc}): {msg}")


def type_info(q_type: int) -> dict:
    t = TypeInfo()
    check(lib().gq_type_info(int(q_type), ctypes.byref(t)), "gq_type_info")
    return {n: getattr(t, n) for n, _ in TypeInfo._fields_}


def prof_enable(tags=None):
    """Enable HIP-event timing for the named kernel tags (None = all, [] = off)."""
    L = lib()
    names = [L.gq_prof_name(i).decode() for i in range(L.gq_prof_ntags())]
    mask = 0
    for i, n in enumerate(names):
        if tags is None or n in tags:
            mask |= 1 << i
    L.gq_prof_enable(mask)
    return names


def prof_collect(busy: bool = False) -> dict:
    """{tag: (total_ms, launches)} since the last collect (synchronises the recorded events); with busy=True
    {tag: (total_ms, launches, busy_ms)}, busy_ms = union of the tag's launch intervals (overlapping launches
    on different streams counted once)."""
    L = lib()
    n = L.gq_prof_ntags()
    ms = (ctypes.c_double * n)()
    cnt = (ctypes.c_long * n)()
    bz = (ctypes.c_double * n)()
    check(L.gq_prof_collect2(ms, cnt, bz), "gq_prof_collect2")
    if busy:
        return {L.gq_prof_name(i).decode(): (ms[i], cnt[i], bz[i]) for i in range(n) if cnt[i]}
    return {L.gq_prof_name(i).decode(): (ms[i], cnt[i]) for i in range(n) if cnt[i]}


# ---- library options (include/gptq_gguf.h lists them): tuning and test switches, process-wide ----
def option_names():
    L = lib()
    return [L.gq_option_name(i).decode() for i in range(L.gq_option_count())]


def option_get(name: str) -> int:
    v = ctypes.c_int64()
    check(lib().gq_option_get(name.encode(), ctypes.byref(v)), "gq_option_get")
    return int(v.value)


def option_default(name: str) -> int:
    v = ctypes.c_int64()
    check(lib().gq_option_default(name.encode(), ctypes.byref(v)), "gq_option_default")
    return int(v.value)


def option_set(name: str, value: int) -> int:
    """-> the previous value."""
    prev = ctypes.c_int64()
    check(lib().gq_option_set(name.encode(), int(value), ctypes.byref(prev)), "gq_option_set")
    return int(prev.value)


class options:
    """`with options(chol_fp32=1, chol_3p_min=0): ...` -- sets library options for the block (None: the default) and restores
    the previous values on exit.  Process-wide: calls enqueued from other threads meanwhile see them too."""

    def __init__(self, **kv):
        self.kv, self.prev = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.prev[k] = option_set(k, option_default(k) if v is None else v)
        return self

    def __exit__(self, *exc):
        for k, v in self.prev.items():
            option_set(k, v)
        return False
