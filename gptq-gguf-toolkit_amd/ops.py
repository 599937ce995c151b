"""Tensor-level entry points: torch CUDA(ROCm) tensors in, C-ABI calls out.

torch is plumbing here (device memory + the current HIP stream); every computation is
a kernel of libgptqgguf_hip.so.  All functions enqueue on torch's current stream and
return without synchronising.
"""
import ctypes
from typing import Optional, Sequence, Tuple

import torch

from . import _cabi
from ._cabi import F16, F32, BF16, Search, check, lib, type_info
from ._cabi import option_get, option_set, options  # noqa: F401  (library tuning / test switches)

_DT = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}


def _stream(t: torch.Tensor):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _cabi.GQError("gptq_gguf_toolkit_amd ops need GPU tensors (no CPU fallback); got a CPU tensor")


def _search(rmin=-1.0, rdelta=0.1, nstep=20, quant_scale="absmax", grid=100, maxshrink=0.8):
    """gq_search_t: make_k_quants' (rmin, rdelta, nstep) and make_quants' quant_scale ("absmax" | "mse", with the
    grid / maxshrink of quant_utils.py:164-191)."""
    mode = getattr(quant_scale, "value", quant_scale)
    if mode not in ("absmax", "mse"):
        raise ValueError(f"quant_scale must be 'absmax' or 'mse', got {quant_scale!r}")
    return ctypes.byref(Search(float(rmin), float(rdelta), int(nstep), int(mode == "mse"), int(grid), float(maxshrink)))


def _idt(q_type):
    return torch.int8 if type_info(q_type)["is_signed"] else torch.uint8


def _u8(t):
    return t.view(torch.uint8) if t.dtype != torch.uint8 else t


def _u16view(t):
    """fp16 tensor -> same storage seen as int16 (bit pattern carrier)."""
    return t.view(torch.int16) if t.dtype == torch.float16 else t


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def workspace_bytes(op: int, R=0, C=0, T=0, block_size=0) -> int:
    return int(lib().gq_workspace_bytes(op, R, C, T, block_size))


def h_accumulate(H: torch.Tensor, X: torch.Tensor, beta: float, alpha: float, ws: Optional[torch.Tensor] = None):
    """H = beta*H + alpha * X^T X in place.  X: [T, C] fp16/bf16/fp32 contiguous (or a list of equal [L, C] blocks)."""
    if isinstance(X, (list, tuple)):
        return h_accumulate_grouped([H], [X], [beta], [alpha], ws)[0]
    _need_cuda(H, X)
    assert H.dtype == torch.float32 and H.is_contiguous() and X.is_contiguous() and X.dim() == 2
    T, C = X.shape
    assert H.shape == (C, C)
    need = workspace_bytes(_cabi.WS_H_ACCUMULATE, 0, C, T)
    if ws is None or ws.numel() < need:
        ws = _ws(need, X.device)
    check(lib().gq_h_accumulate(_ptr(H), _ptr(X), _DT[X.dtype], T, C, beta, alpha, _ptr(ws), ws.numel(), _stream(X)),
          "gq_h_accumulate")
    return H


def h_accumulate_grouped(Hs, Xs, betas, alphas, ws: Optional[torch.Tensor] = None):
    """Up to 8 Hessians in one grid: Hs[i] = betas[i]*Hs[i] + alphas[i] * Xs[i]^T Xs[i].
    Xs[i] is a [T, C] tensor, or a LIST of [L, C] tensors (equal L, contiguous): the per-sample activation
    tensors of the forward hooks, read where they lie (gq_h_accumulate_segments)."""
    n = len(Hs)
    assert n == len(Xs) == len(betas) == len(alphas) and 1 <= n <= 8
    blocks = [list(X) if isinstance(X, (list, tuple)) else [X] for X in Xs]
    _need_cuda(*Hs, *[b for bl in blocks for b in bl])
    dt = blocks[0][0].dtype
    for H, bl in zip(Hs, blocks):
        assert H.dtype == torch.float32 and H.is_contiguous()
        for X in bl:
            assert X.is_contiguous() and X.dim() == 2 and X.dtype == dt and X.shape == bl[0].shape
        assert H.shape == (bl[0].shape[1], bl[0].shape[1])
    Ts = [len(bl) * bl[0].shape[0] for bl in blocks]
    need = sum(workspace_bytes(_cabi.WS_H_ACCUMULATE, 0, bl[0].shape[1], T) for bl, T in zip(blocks, Ts))
    if ws is None or ws.numel() < need:
        ws = _ws(need, blocks[0][0].device)
    vp = ctypes.c_void_p
    Hp = (vp * n)(*[H.data_ptr() for H in Hs])
    Cs = (ctypes.c_int64 * n)(*[bl[0].shape[1] for bl in blocks])
    bs = (ctypes.c_float * n)(*[float(b) for b in betas])
    as_ = (ctypes.c_float * n)(*[float(a) for a in alphas])
    if all(len(bl) == 1 for bl in blocks):
        Xp = (vp * n)(*[bl[0].data_ptr() for bl in blocks])
        Tc = (ctypes.c_int64 * n)(*Ts)
        check(lib().gq_h_accumulate_grouped(n, Hp, Xp, Tc, Cs, bs, as_, _DT[dt], _ptr(ws), ws.numel(),
                                            _stream(blocks[0][0])), "gq_h_accumulate_grouped")
        return Hs
    lists = [(vp * len(bl))(*[X.data_ptr() for X in bl]) for bl in blocks]
    Bp = (vp * n)(*[ctypes.cast(l, vp).value for l in lists])
    nb = (ctypes.c_int64 * n)(*[len(bl) for bl in blocks])
    bt = (ctypes.c_int64 * n)(*[bl[0].shape[0] for bl in blocks])
    check(lib().gq_h_accumulate_segments(n, Hp, Bp, nb, bt, Cs, bs, as_, _DT[dt], _ptr(ws), ws.numel(),
                                         _stream(blocks[0][0])), "gq_h_accumulate_segments")
    return Hs


def h_stage(buf: torch.Tensor, fill: int, x: torch.Tensor):
    """buf[fill : fill + T] = x  (x: [T, C] rows of activations, same dtype as the staging buffer `buf`)."""
    _need_cuda(buf, x)
    assert buf.is_contiguous() and buf.dtype == x.dtype and x.dim() == 2 and x.shape[1] == buf.shape[1]
    assert fill + x.shape[0] <= buf.shape[0]
    if not x.is_contiguous():
        x = x.contiguous()
    row = buf.shape[1] * buf.element_size()
    check(lib().gq_h_stage(ctypes.c_void_p(buf.data_ptr() + fill * row), _ptr(x), x.shape[0] * row, _stream(buf)),
          "gq_h_stage")


def h_stage_many(buf: torch.Tensor, fill: int, xs: Sequence[torch.Tensor]):
    """buf[fill : fill + sum T_k] = cat(xs) in ONE launch (xs: [T_k, C] contiguous blocks of the buffer's dtype, 16-byte
    aligned rows): the staging copies of a whole fold (gq_h_stage_many)."""
    _need_cuda(buf, *xs)
    row = buf.shape[1] * buf.element_size()
    total = sum(x.shape[0] for x in xs)
    assert buf.is_contiguous() and fill + total <= buf.shape[0] and row % 16 == 0
    assert all(x.dtype == buf.dtype and x.dim() == 2 and x.shape[1] == buf.shape[1] and x.is_contiguous() for x in xs)
    n = len(xs)
    srcs = (ctypes.c_void_p * n)(*[x.data_ptr() for x in xs])
    nb = (ctypes.c_int64 * n)(*[x.shape[0] * row for x in xs])
    ws = _ws((2 * n + 1) * 8 + 512, buf.device)
    check(lib().gq_h_stage_many(ctypes.c_void_p(buf.data_ptr() + fill * row), srcs, nb, n, _ptr(ws), ws.numel(), _stream(buf)),
          "gq_h_stage_many")


def h_prepare(H: torch.Tensor, W: torch.Tensor, rel_damp: float, want_flags: bool = False, obq_order: bool = False):
    """In-place dead-channel fix / masking / damping of (H, W); returns (U, not_invertible[int32 tensor])
    (+ col_flags uint8[2*C] if want_flags: the dead / zero-column sets U depends on).  obq_order: damping before
    the zero-column mask (EvoPress FastOBQ, evopress/src/fast_obq.py:133-141, 221-228)."""
    _need_cuda(H, W)
    assert H.dtype == torch.float32 and W.dtype == torch.float32 and H.is_contiguous() and W.is_contiguous()
    R, C = W.shape
    U = torch.empty_like(H)
    flag = torch.empty(1, dtype=torch.int32, device=H.device)  # cleared by the call (hipMemsetAsync on its stream)
    cf = torch.empty(2 * C, dtype=torch.uint8, device=H.device) if want_flags else None
    ws = _ws(workspace_bytes(_cabi.WS_H_PREPARE, R, C), H.device)
    fn = lib().gq_obq_h_prepare if obq_order else lib().gq_h_prepare
    check(fn(_ptr(H), _ptr(W), R, C, rel_damp, _ptr(U), _ptr(flag), _ptr(cf), _ptr(ws), ws.numel(), _stream(H)),
          "gq_obq_h_prepare" if obq_order else "gq_h_prepare")
    return (U, flag, cf) if want_flags else (U, flag)


def w_prepare(col_flags: torch.Tensor, W: torch.Tensor) -> torch.Tensor:
    """Follower of a shared Hessian: zero this W's dead columns; returns mismatch[int32 tensor]
    (0 => the leader's U is this Linear's U)."""
    _need_cuda(col_flags, W)
    assert W.dtype == torch.float32 and W.is_contiguous() and col_flags.numel() == 2 * W.shape[1]
    mm = torch.empty(1, dtype=torch.int32, device=W.device)  # cleared by the call
    check(lib().gq_w_prepare(_ptr(col_flags), _ptr(W), W.shape[0], W.shape[1], _ptr(mm), _stream(W)), "gq_w_prepare")
    return mm


def h_pack_upper(H: torch.Tensor) -> torch.Tensor:
    """The 128x128 tiles of H on and above the diagonal as one contiguous fp32 buffer (the all-reduce payload)."""
    _need_cuda(H)
    C = H.shape[0]
    assert H.dtype == torch.float32 and H.is_contiguous() and H.shape == (C, C) and C % 128 == 0
    nt = C // 128
    buf = torch.empty(nt * (nt + 1) // 2, 128, 128, dtype=torch.float32, device=H.device)
    check(lib().gq_h_pack_upper(_ptr(H), C, _ptr(buf), _stream(H)), "gq_h_pack_upper")
    return buf


def h_unpack_upper(buf: torch.Tensor, H: torch.Tensor) -> torch.Tensor:
    """Inverse of h_pack_upper: writes the tiles back into H and mirrors them below the diagonal."""
    _need_cuda(buf, H)
    C = H.shape[0]
    assert H.dtype == torch.float32 and H.is_contiguous() and buf.dtype == torch.float32 and buf.is_contiguous()
    check(lib().gq_h_unpack_upper(_ptr(buf), C, _ptr(H), _stream(H)), "gq_h_unpack_upper")
    return H


def scale_search(x: torch.Tensor, q_type: int, rmin=-1.0, rdelta=0.1, nstep=20, **mq):
    """get_scale_and_zero on x[rows,256] (row stride free).  Returns (d f16[rows], s[rows,ng], dmin f16[rows], m)."""
    _need_cuda(x)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] == 256 and x.stride(1) == 1
    rows = x.shape[0]
    ng = 256 // type_info(q_type)["group"]
    dev = x.device
    d = torch.empty(rows, dtype=torch.float16, device=dev)
    dmin = torch.empty(rows, dtype=torch.float16, device=dev)
    s = torch.empty(rows, ng, dtype=torch.uint8, device=dev)
    m = torch.empty(rows, ng, dtype=torch.uint8, device=dev)
    check(lib().gq_scale_search(_ptr(x), rows, x.stride(0), int(q_type), _search(rmin, rdelta, nstep, **mq), _ptr(d), 1,
                                _ptr(s), ng, _ptr(dmin), 1, _ptr(m), ng, _stream(x)), "gq_scale_search")
    t = _idt(q_type)
    return d, s.view(t), dmin, m.view(t)


def _alloc_outs(R, C, q_type, dev):
    G = type_info(q_type)["group"]
    return (torch.empty(R, C, dtype=torch.uint8, device=dev), torch.empty(R, C // 256, dtype=torch.float16, device=dev),
            torch.empty(R, C // G, dtype=torch.uint8, device=dev),
            torch.empty(R, C // 256, dtype=torch.float16, device=dev),
            torch.empty(R, C // G, dtype=torch.uint8, device=dev))


_side_streams = {}


def _streams(device, n):
    pool = _side_streams.setdefault(device, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device))
    return pool[:n]


def gptq_quantize(W: torch.Tensor, U: torch.Tensor, q_type: int, block_size=128, static_groups=False, rmin=-1.0,
                  rdelta=0.1, nstep=20, ws: Optional[torch.Tensor] = None, row_chunks: Optional[int] = None,
                  row_ends: Optional[Sequence[int]] = None, panel_researches: Optional[torch.Tensor] = None, **mq):
    """GPTQ.step body.  W (fp32, contiguous) is updated IN PLACE to the dequantized matrix.
    Returns (qweight, d, s, dmin, m).

    `panel_researches` (gq_gptq_quantize_slice): W is a ROW SLICE of a matrix several ranks quantize together; the int32
    device tensor receives the number of panel-wide re-searches of the slice's scale searches (0: the slice's rows equal
    the whole matrix's rows bit for bit -- quant_utils.py:250-252 is the one place the reference looks across all rows).

    `row_ends` (gq_gptq_quantize_stacked): W holds several Linears that share U, one under the other -- matrix k is rows
    [row_ends[k-1], row_ends[k]), multiples of 64, the last one == R.  One walk over the columns instead of one per
    Linear; every output row equals the separate calls bit for bit (the scale search's panel-wide `continue`,
    quant_utils.py:250-252, is evaluated per stacked matrix).

    Rows of W are independent given U (gptq.py:222-270 never mixes rows), so a tall matrix is cut into
    `row_chunks` contiguous row ranges, each a separate gq_gptq_quantize call on its own HIP stream: the
    latency-bound column-loop kernel of one chunk overlaps with the trailing-update GEMM of another.
    Results are identical to one call.  Default 1 (what BlockSchedule uses): every extra chunk multiplies the
    launch count; measured no gain inside a block's four-chain schedule."""
    _need_cuda(W, U)
    assert W.dtype == torch.float32 and U.dtype == torch.float32 and W.is_contiguous() and U.is_contiguous()
    R, C = W.shape
    q, d, s, dmin, m = _alloc_outs(R, C, q_type, W.device)
    bs = int(block_size or 0)
    if row_chunks is None:
        row_chunks = 1
    if row_chunks > 1 and (R % (64 * row_chunks) or ws is not None):
        row_chunks = 1

    if row_ends is not None and len(row_ends) > 1:
        ends = (ctypes.c_int64 * len(row_ends))(*[int(e) for e in row_ends])
        need = workspace_bytes(_cabi.WS_GPTQ_QUANTIZE, R, C, 0, bs)
        if ws is None or ws.numel() < need:
            ws = _ws(need, W.device)
        check(lib().gq_gptq_quantize_stacked(_ptr(W), _ptr(U), R, C, int(q_type), bs, int(bool(static_groups)),
                                             _search(rmin, rdelta, nstep, **mq), _ptr(q), _ptr(d), _ptr(s), _ptr(dmin),
                                             _ptr(m), ends, len(row_ends), _ptr(ws), ws.numel(), _stream(W)),
              "gq_gptq_quantize_stacked")
        t = _idt(q_type)
        return q.view(t), d, s.view(t), dmin, m.view(t)

    if panel_researches is not None:
        assert panel_researches.dtype == torch.int32 and panel_researches.is_cuda and panel_researches.numel() >= 1
        need = workspace_bytes(_cabi.WS_GPTQ_QUANTIZE, R, C, 0, bs)
        if ws is None or ws.numel() < need:
            ws = _ws(need, W.device)
        check(lib().gq_gptq_quantize_slice(_ptr(W), _ptr(U), R, C, int(q_type), bs, int(bool(static_groups)),
                                           _search(rmin, rdelta, nstep, **mq), _ptr(q), _ptr(d), _ptr(s), _ptr(dmin), _ptr(m),
                                           _ptr(panel_researches), _ptr(ws), ws.numel(), _stream(W)), "gq_gptq_quantize_slice")
        t = _idt(q_type)
        return q.view(t), d, s.view(t), dmin, m.view(t)

    def one(r0, r1, wsbuf):
        n = r1 - r0
        need = workspace_bytes(_cabi.WS_GPTQ_QUANTIZE, n, C, 0, bs)
        if wsbuf is None or wsbuf.numel() < need:
            wsbuf = _ws(need, W.device)
        check(lib().gq_gptq_quantize(_ptr(W[r0:r1]), _ptr(U), n, C, int(q_type), bs, int(bool(static_groups)),
                                     _search(rmin, rdelta, nstep, **mq), _ptr(q[r0:r1]), _ptr(d[r0:r1]), _ptr(s[r0:r1]),
                                     _ptr(dmin[r0:r1]), _ptr(m[r0:r1]), _ptr(wsbuf), wsbuf.numel(), _stream(W)),
              "gq_gptq_quantize")
        return wsbuf

    if row_chunks == 1:
        one(0, R, ws)
    else:
        cur = torch.cuda.current_stream(W.device)
        fork = torch.cuda.Event()
        fork.record(cur)
        step = R // row_chunks
        keep = []
        for k, st in enumerate(_streams(W.device, row_chunks)):
            st.wait_event(fork)
            with torch.cuda.stream(st):
                keep.append(one(k * step, (k + 1) * step, None))
            ev = torch.cuda.Event()
            ev.record(st)
            cur.wait_event(ev)
        for b in keep:  # scratch was allocated on side streams; it is dead once `cur` has passed the joins
            b.record_stream(cur)
    t = _idt(q_type)
    return q.view(t), d, s.view(t), dmin, m.view(t)


def uses_helper_stream(R: int, C: int, block_size) -> bool:
    """Will the column loop of an R x C matrix run its far updates on the library's helper stream (gq_gptq_uses_helper_stream)?"""
    return bool(lib().gq_gptq_uses_helper_stream(int(R), int(C), int(block_size or 0)))


def far_helper_enable(on: bool) -> bool:
    """Allow / forbid the library's helper stream for the column loops enqueued from now on; returns the previous setting."""
    return bool(lib().gq_far_helper_enable(int(bool(on))))


def gptq_quantize_perm(W: torch.Tensor, U: torch.Tensor, q_type: int, perm: torch.Tensor, d, s, dmin, m, block_size=128,
                       ws: Optional[torch.Tensor] = None) -> torch.Tensor:
    """GPTQ.step body with act_order (gptq.py:208-216, 233-235).  W (fp32) and U are already permuted by `perm`
    (int32 [C], original column of each position); (d, s, dmin, m) are the static scales of the ORIGINAL column
    groups.  W becomes the dequantized matrix (permuted); returns qweight in permuted positions."""
    _need_cuda(W, U, perm, d, s, dmin, m)
    assert W.dtype == torch.float32 and U.dtype == torch.float32 and W.is_contiguous() and U.is_contiguous()
    assert perm.dtype == torch.int32 and perm.is_contiguous() and perm.numel() == W.shape[1]
    R, C = W.shape
    q = torch.empty(R, C, dtype=torch.uint8, device=W.device)
    bs = int(block_size or 0)
    need = workspace_bytes(_cabi.WS_GPTQ_QUANTIZE, R, C, 0, bs)
    if ws is None or ws.numel() < need:
        ws = _ws(need, W.device)
    check(lib().gq_gptq_quantize_perm(_ptr(W), _ptr(U), R, C, int(q_type), bs, _ptr(perm), _ptr(d.contiguous()),
                                      _ptr(s.contiguous()), _ptr(dmin.contiguous()), _ptr(m.contiguous()), _ptr(q),
                                      _ptr(ws), ws.numel(), _stream(W)), "gq_gptq_quantize_perm")
    return q.view(_idt(q_type))


def obq_quantize(W: torch.Tensor, U: torch.Tensor, bits: int, group_size: int = 0, sym: bool = False, block_size=128,
                 ws: Optional[torch.Tensor] = None):
    """EvoPress FastOBQ.step for one bit width (evopress/src/fast_obq.py:146-200).  W (fp32, contiguous) becomes
    the dequantized matrix.  Returns (qweight u8 [R, C], scale f32 [R, C/G], zero f32 [R, C/G])."""
    _need_cuda(W, U)
    assert W.dtype == torch.float32 and U.dtype == torch.float32 and W.is_contiguous() and U.is_contiguous()
    R, C = W.shape
    ng = C // group_size if group_size else 1
    q = torch.empty(R, C, dtype=torch.uint8, device=W.device)
    scale = torch.empty(R, ng, dtype=torch.float32, device=W.device)
    zero = torch.empty(R, ng, dtype=torch.float32, device=W.device)
    bs = int(block_size or 0)
    need = workspace_bytes(_cabi.WS_GPTQ_QUANTIZE, R, C, 0, bs)
    if ws is None or ws.numel() < need:
        ws = _ws(need, W.device)
    check(lib().gq_obq_quantize(_ptr(W), _ptr(U), R, C, int(bits), int(group_size or 0), int(bool(sym)), bs, _ptr(q),
                                _ptr(scale), _ptr(zero), _ptr(ws), ws.numel(), _stream(W)), "gq_obq_quantize")
    return q, scale, zero


def rtn_quantize(W: torch.Tensor, q_type: int, rmin=-1.0, rdelta=0.1, nstep=20, **mq):
    _need_cuda(W)
    assert W.is_contiguous() and W.dim() == 2 and W.dtype in _DT
    R, C = W.shape
    q, d, s, dmin, m = _alloc_outs(R, C, q_type, W.device)
    check(lib().gq_rtn_quantize(_ptr(W), _DT[W.dtype], R, C, int(q_type), _search(rmin, rdelta, nstep, **mq), _ptr(q),
                                _ptr(d), _ptr(s), _ptr(dmin), _ptr(m), _stream(W)), "gq_rtn_quantize")
    t = _idt(q_type)
    return q.view(t), d, s.view(t), dmin, m.view(t)


def dequantize(q_type: int, q, d, s, dmin, m, out_dtype=torch.float32) -> torch.Tensor:
    _need_cuda(q, d, s, dmin, m)
    R, C = q.shape
    out = torch.empty(R, C, dtype=out_dtype, device=q.device)
    check(lib().gq_dequantize(int(q_type), _ptr(q.contiguous()), _ptr(d.contiguous()), _ptr(s.contiguous()),
                              _ptr(dmin.contiguous()), _ptr(m.contiguous()), R, C, _ptr(out), _DT[out_dtype],
                              _stream(q)), "gq_dequantize")
    return out


def pack(q_type: int, q, d, s, dmin=None, m=None) -> torch.Tensor:
    """-> uint8 [R, C/256*type_size] on the device.  Inputs are not modified."""
    _need_cuda(q, d, s, dmin, m)
    R, C = q.shape
    ts = type_info(q_type)["type_size"]
    out = torch.empty(R, C // 256 * ts, dtype=torch.uint8, device=q.device)
    check(lib().gq_pack(int(q_type), _ptr(q.contiguous()), _ptr(d.contiguous()), _ptr(s.contiguous()),
                        _ptr(dmin.contiguous() if dmin is not None else None),
                        _ptr(m.contiguous() if m is not None else None), R, C, _ptr(out), _stream(q)), "gq_pack")
    return out


def trailing_update(Cm: torch.Tensor, A: torch.Tensor, B: torch.Tensor):
    """Cm -= A @ B (fp32; k-ordered fma chain then one subtraction per element)."""
    _need_cuda(Cm, A, B)
    M, K = A.shape
    K2, N = B.shape
    assert K == K2 and Cm.shape == (M, N) and Cm.stride(1) == 1 and A.stride(1) == 1 and B.stride(1) == 1
    check(lib().gq_trailing_update(_ptr(Cm), Cm.stride(0), _ptr(A), A.stride(0), _ptr(B), B.stride(0), M, N, K,
                                   _stream(Cm)), "gq_trailing_update")
    return Cm


def chol_gemm(Cm: torch.Tensor, A: torch.Tensor, B: torch.Tensor, trans_b: bool, mode: int, k_range: int = 0,
              lower: bool = False, planes: int = 3):
    """One product of the blocked Cholesky chain through the pre-split image kernels (gq_chol_gemm):
    mode 0: Cm -= A op(B), 1: Cm = A op(B), 2: Cm = -(A op(B)); fp32-GEMM accuracy (tolerance class)."""
    _need_cuda(Cm, A, B)
    M, K = A.shape
    N = B.shape[0] if trans_b else B.shape[1]
    assert (B.shape[1] if trans_b else B.shape[0]) == K and Cm.shape == (M, N)
    assert Cm.stride(1) == 1 and A.stride(1) == 1 and B.stride(1) == 1
    ws = _ws(workspace_bytes(_cabi.WS_CHOL_GEMM, M, N, K), Cm.device)
    check(lib().gq_chol_gemm(_ptr(Cm), Cm.stride(0), _ptr(A), A.stride(0), _ptr(B), B.stride(0), M, N, K, int(trans_b),
                             int(mode), int(k_range), int(lower), int(planes), _ptr(ws), ws.numel(), _stream(Cm)),
          "gq_chol_gemm")
    return Cm


def stage_to_host(dst: torch.Tensor, src: torch.Tensor, stream: "torch.cuda.Stream") -> None:
    """src (device, contiguous) -> dst (pinned host memory, same byte size) by a copy kernel on `stream`
    (gq_stage_to_host: no hipMemcpy, no copy-engine lock shared with the launching thread)."""
    _need_cuda(src)
    assert src.is_contiguous() and dst.is_contiguous() and not dst.is_cuda
    n = src.numel() * src.element_size()
    assert dst.numel() * dst.element_size() == n
    check(lib().gq_stage_to_host(_ptr(dst), _ptr(src), n, ctypes.c_void_p(stream.cuda_stream)), "gq_stage_to_host")


def fwd_rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """LlamaRMSNorm.forward in one pass (gq_fwd_rmsnorm); x [..., C] fp16 / bf16 contiguous, weight [C] of the same dtype."""
    _need_cuda(x, weight)
    assert x.is_contiguous() and weight.is_contiguous() and weight.dtype == x.dtype and weight.numel() == x.shape[-1]
    out = torch.empty_like(x)
    C = x.shape[-1]
    check(lib().gq_fwd_rmsnorm(_ptr(x), _ptr(weight), _ptr(out), x.numel() // C, C, float(eps), _DT[x.dtype], _stream(x)),
          "gq_fwd_rmsnorm")
    return out


def fwd_rmsnorm_ordered(x: torch.Tensor, weight: torch.Tensor, eps: float, want_stats: bool = False):
    """fwd_rmsnorm with the statistics summed in ATen's order (gq_fwd_rmsnorm_ordered; C % 512 == 0).  want_stats: also
    return fp32 [rows, 2] = (mean(x^2), rsqrt(mean + eps)) as the kernel computed them."""
    _need_cuda(x, weight)
    assert x.is_contiguous() and weight.is_contiguous() and weight.dtype == x.dtype and weight.numel() == x.shape[-1]
    out = torch.empty_like(x)
    C = x.shape[-1]
    stats = torch.empty(x.numel() // C, 2, dtype=torch.float32, device=x.device) if want_stats else None
    check(lib().gq_fwd_rmsnorm_ordered(_ptr(x), _ptr(weight), _ptr(out), x.numel() // C, C, float(eps), _DT[x.dtype],
                                       _ptr(stats), _stream(x)), "gq_fwd_rmsnorm_ordered")
    return (out, stats) if want_stats else out


def fwd_rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """apply_rotary_pos_emb on one projection (gq_fwd_rope): x [B, L, H, D] contiguous (q_proj(h).view(B, L, H, D)),
    cos / sin [B, L, D] contiguous, same dtype.  Returns [B, L, H, D]."""
    _need_cuda(x, cos, sin)
    B, L, H, D = x.shape
    assert x.is_contiguous() and cos.is_contiguous() and sin.is_contiguous() and cos.dtype == x.dtype == sin.dtype
    assert cos.shape == (B, L, D) and sin.shape == (B, L, D)
    out = torch.empty_like(x)
    check(lib().gq_fwd_rope(_ptr(x), _ptr(cos), _ptr(sin), _ptr(out), B * L, H, D, _DT[x.dtype], _stream(x)), "gq_fwd_rope")
    return out


def fwd_silu_mul(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    """silu(gate) * up (gq_fwd_silu_mul), fp16 / bf16 contiguous tensors of one shape."""
    _need_cuda(gate, up)
    assert gate.is_contiguous() and up.is_contiguous() and gate.shape == up.shape and gate.dtype == up.dtype
    out = torch.empty_like(gate)
    check(lib().gq_fwd_silu_mul(_ptr(gate), _ptr(up), _ptr(out), gate.numel(), _DT[gate.dtype], _stream(gate)),
          "gq_fwd_silu_mul")
    return out


def group_search(x: torch.Tensor, q_type: int, rmin=-1.0, rdelta=0.1, nstep=20, **mq):
    """make_k_quants / make_quants on a [rows,256] panel (fp32, or fp16/bf16 with per-op rounding).
    Returns (group_scale f32[rows,ng], group_zero f32[rows,ng], d, s, dmin, m)."""
    _need_cuda(x)
    assert x.dim() == 2 and x.shape[1] == 256 and x.stride(1) == 1 and x.dtype in _DT
    rows = x.shape[0]
    ng = 256 // type_info(q_type)["group"]
    dev = x.device
    gs = torch.empty(rows, ng, dtype=torch.float32, device=dev)
    gz = torch.empty(rows, ng, dtype=torch.float32, device=dev)
    d = torch.empty(rows, dtype=torch.float16, device=dev)
    dmin = torch.empty(rows, dtype=torch.float16, device=dev)
    s = torch.empty(rows, ng, dtype=torch.uint8, device=dev)
    m = torch.empty(rows, ng, dtype=torch.uint8, device=dev)
    check(lib().gq_group_search(_ptr(x), _DT[x.dtype], rows, x.stride(0), int(q_type), _search(rmin, rdelta, nstep, **mq),
                                _ptr(gs), _ptr(gz), _ptr(d), _ptr(s), _ptr(dmin), _ptr(m), _stream(x)),
          "gq_group_search")
    t = _idt(q_type)
    return gs, gz, d, s.view(t), dmin, m.view(t)
