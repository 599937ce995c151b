"""ctypes view of the CPU oracle (oracle/libgq_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
numpy in / numpy out; fp16 fields are carried as np.uint16 bit patterns (use
.view(np.float16) to look at them).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgq_oracle.so")

Q2_K, Q3_K, Q4_K, Q5_K, Q6_K = 10, 11, 12, 13, 14
ALL_TYPES = (Q2_K, Q3_K, Q4_K, Q5_K, Q6_K)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "gq_oracle.c")
    hdr = os.path.join(_HERE, "gq_oracle.h")
    stale = (not os.path.exists(_SO)) or any(
        os.path.getmtime(p) > os.path.getmtime(_SO) for p in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libgq_oracle.so"])
    return _SO


class _TI(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in
                ("bits", "qmin", "qmax", "scale_maxq", "group", "is_signed", "k_search", "type_size")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        vp, i64, ci, cd, cf = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_float
        L.gqo_type_info.argtypes = [ci, ctypes.POINTER(_TI)]
        L.gqo_aten_sum.argtypes = [vp, ci]
        L.gqo_aten_sum.restype = cf
        L.gqo_f32_to_f16.argtypes = [cf]
        L.gqo_f32_to_f16.restype = ctypes.c_uint16
        L.gqo_make_k_quants.argtypes = [vp, i64, ci, ci, cd, cd, ci, vp, vp]
        L.gqo_make_k_quants.restype = None
        L.gqo_make_quants.argtypes = [vp, i64, ci, ci, vp, vp]
        L.gqo_make_quants.restype = None
        L.gqo_scale_search.argtypes = [vp, i64, i64, ci, cd, cd, ci, vp, i64, vp, i64, vp, i64, vp, i64]
        L.gqo_scale_search.restype = None
        L.gqo_gptq_step.argtypes = [vp, vp, i64, i64, ci, ci, ci, cd, cd, ci, vp, vp, vp, vp, vp]
        L.gqo_gptq_step_perm.argtypes = [vp, vp, i64, i64, ci, ci, vp, vp, vp, vp, vp, vp]
        L.gqo_gptq_step_perm.restype = None
        L.gqo_gptq_step.restype = None
        L.gqo_rtn_quantize.argtypes = [vp, i64, i64, ci, cd, cd, ci, vp, vp, vp, vp, vp]
        L.gqo_rtn_quantize.restype = None
        L.gqo_rtn_quantize_lp.argtypes = [vp, ci, i64, i64, ci, cd, cd, ci, vp, vp, vp, vp, vp]
        L.gqo_rtn_quantize_lp.restype = None
        L.gqo_dequantize.argtypes = [ci, vp, vp, vp, vp, vp, i64, i64, vp]
        L.gqo_dequantize.restype = None
        L.gqo_pack.argtypes = [ci, vp, vp, vp, vp, vp, i64, i64, vp]
        L.gqo_h_accumulate.argtypes = [vp, vp, i64, i64, cf, cf]
        L.gqo_h_accumulate.restype = None
        L.gqo_h_prepare.argtypes = [vp, vp, i64, i64, cf, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def type_info(q_type: int) -> dict:
    t = _TI()
    if lib().gqo_type_info(int(q_type), ctypes.byref(t)):
        raise ValueError(f"unsupported q_type {q_type}")
    return {n: getattr(t, n) for n, _ in _TI._fields_}


def _idt(q_type):
    return np.int8 if type_info(q_type)["is_signed"] else np.uint8


def aten_sum(v: np.ndarray) -> np.float32:
    v = np.ascontiguousarray(v, np.float32)
    return np.float32(lib().gqo_aten_sum(_p(v), v.size))


def make_k_quants(x, bits, rmin=-1.0, rdelta=0.1, nstep=20):
    x = np.ascontiguousarray(x, np.float32)
    n, G = x.shape
    sc, ze = np.empty(n, np.float32), np.empty(n, np.float32)
    lib().gqo_make_k_quants(_p(x), n, G, bits, rmin, rdelta, nstep, _p(sc), _p(ze))
    return sc, ze


def obq_step(W, U, bits, group_size=128, sym=False, block_size=128):
    """EvoPress FastOBQ.step for one bit width (evopress/src/fast_obq.py:131-200) given U.
    Returns (W_dequantized, qweight u8 [R, C], scale f32 [R, C/G], zero f32 [R, C/G])."""
    L = lib()
    vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    L.gqo_obq_step.argtypes = [vp, vp, i64, i64, ci, ci, ci, ci, vp, vp, vp]
    L.gqo_obq_step.restype = None
    W = np.array(W, np.float32, order="C", copy=True)
    U = np.ascontiguousarray(U, np.float32)
    R, C = W.shape
    ng = C // group_size if group_size else 1
    q = np.empty((R, C), np.uint8)
    sc, ze = np.empty((R, ng), np.float32), np.empty((R, ng), np.float32)
    L.gqo_obq_step(_p(W), _p(U), R, C, bits, group_size or 0, int(sym), block_size or 0, _p(q), _p(sc), _p(ze))
    return W, q, sc, ze


def set_quant_scale(mode="absmax", grid=100, maxshrink=0.8):
    """make_quants' quant_scale (quant_utils.py:164-191) for every later call of this module: "absmax" or "mse"."""
    L = lib()
    L.gqo_set_quant_scale.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double]
    L.gqo_set_quant_scale.restype = None
    L.gqo_set_quant_scale(int(mode == "mse"), int(grid), float(maxshrink))


def make_quants(x, bits):
    x = np.ascontiguousarray(x, np.float32)
    n, G = x.shape
    sc, ze = np.empty(n, np.float32), np.empty(n, np.float32)
    lib().gqo_make_quants(_p(x), n, G, bits, _p(sc), _p(ze))
    return sc, ze


def scale_search(x, q_type, rmin=-1.0, rdelta=0.1, nstep=20):
    """get_scale_and_zero on x[rows,256] -> (d u16[rows], s[rows,ng], dmin u16[rows], m[rows,ng])"""
    x = np.ascontiguousarray(x, np.float32)
    rows, w = x.shape
    assert w == 256
    ng = 256 // type_info(q_type)["group"]
    d, dmin = np.empty(rows, np.uint16), np.empty(rows, np.uint16)
    s, m = np.empty((rows, ng), np.uint8), np.empty((rows, ng), np.uint8)
    lib().gqo_scale_search(_p(x), rows, 256, q_type, rmin, rdelta, nstep,
                           _p(d), 1, _p(s), ng, _p(dmin), 1, _p(m), ng)
    return d, s.view(_idt(q_type)), dmin, m.view(_idt(q_type))


def _alloc_outs(R, C, q_type):
    G = type_info(q_type)["group"]
    return (np.empty((R, C), np.uint8), np.empty((R, C // 256), np.uint16),
            np.empty((R, C // G), np.uint8), np.empty((R, C // 256), np.uint16),
            np.empty((R, C // G), np.uint8))


def panel_researches(reset=True) -> int:
    """Skipped search iterations (quant_utils.py:251-252) whose candidate some group would have taken, since the last
    reset -- what gq_gptq_quantize_slice reports for a row slice."""
    L = lib()
    L.gqo_panel_researches.restype = ctypes.c_int64
    return int(L.gqo_panel_researches(int(bool(reset))))


def gptq_step(W, U, q_type, block_size=128, static_groups=False, rmin=-1.0, rdelta=0.1, nstep=20):
    """GPTQ.step.  Returns (W_dequantized, qweight, d, s, dmin, m) -- d/dmin as uint16 bits."""
    W = np.array(W, np.float32, order="C", copy=True)
    U = np.ascontiguousarray(U, np.float32)
    R, C = W.shape
    q, d, s, dmin, m = _alloc_outs(R, C, q_type)
    lib().gqo_gptq_step(_p(W), _p(U), R, C, q_type, block_size or 0, int(static_groups),
                        rmin, rdelta, nstep, _p(q), _p(d), _p(s), _p(dmin), _p(m))
    t = _idt(q_type)
    return W, q.view(t), d, s.view(t), dmin, m.view(t)


def gptq_step_perm(W, U, q_type, perm, d, s, dmin, m, block_size=128):
    """GPTQ.step with act_order: W/U permuted, (d, s, dmin, m) = static scales of the original groups.
    Returns (W_dequantized_permuted, qweight_in_permuted_positions)."""
    W = np.array(W, np.float32, order="C", copy=True)
    U = np.ascontiguousarray(U, np.float32)
    R, C = W.shape
    perm = np.ascontiguousarray(perm, np.int32)
    q = np.zeros((R, C), np.uint8)
    d = np.ascontiguousarray(d).view(np.uint16)
    dmin = np.ascontiguousarray(dmin).view(np.uint16)
    s8 = np.ascontiguousarray(s).view(np.uint8)
    m8 = np.ascontiguousarray(m).view(np.uint8)
    lib().gqo_gptq_step_perm(_p(W), _p(U), R, C, q_type, block_size or 0, _p(perm), _p(d), _p(s8), _p(dmin), _p(m8), _p(q))
    return W, q.view(_idt(q_type))


def rtn_quantize(W, q_type, rmin=-1.0, rdelta=0.1, nstep=20):
    W = np.ascontiguousarray(W, np.float32)
    R, C = W.shape
    q, d, s, dmin, m = _alloc_outs(R, C, q_type)
    lib().gqo_rtn_quantize(_p(W), R, C, q_type, rmin, rdelta, nstep, _p(q), _p(d), _p(s), _p(dmin), _p(m))
    t = _idt(q_type)
    return q.view(t), d, s.view(t), dmin, m.view(t)


def rtn_quantize_lp(W, rmode, q_type, rmin=-1.0, rdelta=0.1, nstep=20):
    """RTN with make_*quants emulated in fp16 (rmode=1) / bf16 (rmode=2); W = weight values as fp32."""
    W = np.ascontiguousarray(W, np.float32)
    R, C = W.shape
    q, d, s, dmin, m = _alloc_outs(R, C, q_type)
    lib().gqo_rtn_quantize_lp(_p(W), rmode, R, C, q_type, rmin, rdelta, nstep, _p(q), _p(d), _p(s), _p(dmin), _p(m))
    t = _idt(q_type)
    return q.view(t), d, s.view(t), dmin, m.view(t)


def _as_u8(a):
    return np.ascontiguousarray(a).view(np.uint8)


def dequantize(q_type, q, d, s, dmin, m):
    R, C = q.shape
    out = np.empty((R, C), np.float32)
    lib().gqo_dequantize(q_type, _p(_as_u8(q)), _p(np.ascontiguousarray(d).view(np.uint16)), _p(_as_u8(s)),
                         _p(np.ascontiguousarray(dmin).view(np.uint16)), _p(_as_u8(m)), R, C, _p(out))
    return out


def pack(q_type, q, d, s, dmin=None, m=None):
    R, C = q.shape
    ti = type_info(q_type)
    out = np.empty((R, C // 256 * ti["type_size"]), np.uint8)
    if dmin is None:
        dmin = np.zeros((R, C // 256), np.uint16)
        m = np.zeros((R, C // ti["group"]), np.uint8)
    rc = lib().gqo_pack(q_type, _p(_as_u8(q)), _p(np.ascontiguousarray(d).view(np.uint16)), _p(_as_u8(s)),
                        _p(np.ascontiguousarray(dmin).view(np.uint16)), _p(_as_u8(m)), R, C, _p(out))
    if rc:
        raise ValueError(f"gqo_pack failed rc={rc}")
    return out


def h_accumulate(H, X, beta, alpha):
    H = np.array(H, np.float32, order="C", copy=True)
    X = np.ascontiguousarray(X, np.float32)
    T, C = X.shape
    lib().gqo_h_accumulate(_p(H), _p(X), T, C, beta, alpha)
    return H


def h_prepare(H, W, rel_damp=0.01, obq_order=False):
    """Returns (U, H_mutated, W_mutated, not_invertible).  obq_order: EvoPress FastOBQ's damp-then-mask."""
    H = np.array(H, np.float32, order="C", copy=True)
    W = np.array(W, np.float32, order="C", copy=True)
    R, C = W.shape
    U = np.empty((C, C), np.float32)
    fn = lib().gqo_obq_h_prepare if obq_order else lib().gqo_h_prepare
    fn.argtypes = lib().gqo_h_prepare.argtypes
    fn.restype = ctypes.c_int
    bad = fn(_p(H), _p(W), R, C, rel_damp, _p(U))
    return U, H, W, bool(bad)


def make_quants_lp(x, q_type, rmode, rmin=-1.0, rdelta=0.1, nstep=20):
    """make_k_quants / make_quants on x[n,G] with fp16 (rmode=1) / bf16 (rmode=2) / fp32 (0) per-op rounding."""
    ti = type_info(q_type)
    x = np.ascontiguousarray(x, np.float32)
    n, G = x.shape
    assert G == ti["group"]
    sc, ze = np.empty(n, np.float32), np.empty(n, np.float32)
    L = lib()
    L.gqo_make_quants_lp.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_void_p,
                                     ctypes.c_void_p]
    L.gqo_make_quants_lp.restype = None
    L.gqo_make_quants_lp(_p(x), n, G, ti["bits"], ti["k_search"], rmode, rmin, rdelta, nstep, _p(sc), _p(ze))
    return sc, ze
