"""CPU stand-in for gptq_gguf_toolkit_amd.ops, backed by the ORACLE, used ONLY by the CPU tests of
the host logic (driver walk, Hessian sharing, 2-rank gloo exchange).  It mirrors the ops signatures
on CPU torch tensors.  This is test scaffolding: the product package never imports it."""
import numpy as np
import torch

from oracle import oracle as O

calls = {"h_accumulate": 0, "h_prepare": 0, "w_prepare": 0, "gptq_quantize": 0, "gptq_quantize_stacked": 0,
         "gptq_quantize_slice": 0}


def _f16(bits):
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).view(torch.float16)


def _bits(t):
    return t.contiguous().view(torch.int16).numpy().view(np.uint16)


def h_accumulate(H, X, beta, alpha, ws=None):
    calls["h_accumulate"] += 1
    if isinstance(X, (list, tuple)):
        X = torch.cat(list(X))
    H.copy_(torch.from_numpy(O.h_accumulate(H.numpy(), X.float().numpy(), beta, alpha)))
    return H


def h_stage(buf, fill, x):
    buf[fill:fill + x.shape[0]].copy_(x)


def h_stage_many(buf, fill, xs):
    for x in xs:
        buf[fill:fill + x.shape[0]] = x
        fill += x.shape[0]


def h_accumulate_grouped(Hs, Xs, betas, alphas, ws=None):
    calls["h_accumulate_grouped"] = calls.get("h_accumulate_grouped", 0) + 1
    for H, X, b, a in zip(Hs, Xs, betas, alphas):
        h_accumulate(H, X, b, a)
    return Hs


def h_prepare(H, W, rel_damp, want_flags=False, obq_order=False):
    calls["h_prepare"] += 1
    H0 = H.numpy().copy()
    dead = (np.diag(H0) == 0)
    U, H2, W2, bad = O.h_prepare(H0, W.numpy(), rel_damp, obq_order=obq_order)
    H.copy_(torch.from_numpy(H2))
    W.copy_(torch.from_numpy(W2))
    flag = torch.tensor([int(bad)], dtype=torch.int32)
    zc = dead | (W2 == 0).all(axis=0)
    cf = torch.from_numpy(np.concatenate([dead, zc]).astype(np.uint8))
    U = torch.from_numpy(U)
    return (U, flag, cf) if want_flags else (U, flag)


def w_prepare(cf, W):
    calls["w_prepare"] += 1
    C = W.shape[1]
    dead, zc = cf[:C].bool(), cf[C:].bool()
    W[:, dead] = 0
    mine = dead | (W == 0).all(dim=0)
    return torch.tensor([int(not torch.equal(mine, zc))], dtype=torch.int32)


def _mode(mq):
    O.set_quant_scale(mq.get("quant_scale", "absmax"), mq.get("grid", 100), mq.get("maxshrink", 0.8))


def gptq_quantize(W, U, q_type, block_size=128, static_groups=False, rmin=-1.0, rdelta=0.1, nstep=20, ws=None,
                  row_ends=None, panel_researches=None, **mq):
    calls["gptq_quantize"] += 1
    if row_ends is not None and len(row_ends) > 1:
        # row-stacked Linears that share U (gq_gptq_quantize_stacked): the oracle, matrix by matrix
        calls["gptq_quantize_stacked"] += 1
        assert row_ends[-1] == W.shape[0] and all(e % 64 == 0 for e in row_ends)
        parts, r0 = [], 0
        for r1 in row_ends:
            calls["gptq_quantize"] -= 1
            parts.append(gptq_quantize(W[r0:r1], U, q_type, block_size, static_groups, rmin, rdelta, nstep, **mq))
            r0 = r1
        return tuple(torch.cat([p[i] for p in parts]) for i in range(5))
    _mode(mq)
    O.panel_researches(reset=True)
    Wd, q, d, s, dmin, m = O.gptq_step(W.numpy(), U.numpy(), q_type, block_size, static_groups, rmin, rdelta, nstep)
    if panel_researches is not None:  # gq_gptq_quantize_slice
        calls["gptq_quantize_slice"] += 1
        panel_researches.fill_(O.panel_researches(reset=True))
    _mode({})
    W.copy_(torch.from_numpy(Wd))
    return torch.from_numpy(q), _f16(d), torch.from_numpy(s), _f16(dmin), torch.from_numpy(m)


def uses_helper_stream(R, C, block_size):
    return False


def far_helper_enable(on):
    return True


def gptq_quantize_perm(W, U, q_type, perm, d, s, dmin, m, block_size=128, ws=None):
    calls["gptq_quantize"] += 1
    Wd, q = O.gptq_step_perm(W.numpy(), U.numpy(), q_type, perm.numpy(), _bits(d), s.numpy(), _bits(dmin), m.numpy(), block_size)
    W.copy_(torch.from_numpy(Wd))
    return torch.from_numpy(q)


def obq_quantize(W, U, bits, group_size=0, sym=False, block_size=128, ws=None):
    Wd, q, sc, ze = O.obq_step(W.numpy(), U.numpy(), bits, group_size or 0, sym, block_size or 0)
    W.copy_(torch.from_numpy(Wd))
    return torch.from_numpy(q), torch.from_numpy(sc), torch.from_numpy(ze)


def rtn_quantize(W, q_type, rmin=-1.0, rdelta=0.1, nstep=20, **mq):
    _mode(mq)
    q, d, s, dmin, m = O.rtn_quantize(W.float().numpy(), q_type, rmin, rdelta, nstep)
    _mode({})
    return torch.from_numpy(q), _f16(d), torch.from_numpy(s), _f16(dmin), torch.from_numpy(m)


def dequantize(q_type, q, d, s, dmin, m, out_dtype=torch.float32):
    w = O.dequantize(q_type, q.numpy(), _bits(d), s.numpy(), _bits(dmin), m.numpy())
    return torch.from_numpy(w).to(out_dtype)


def pack(q_type, q, d, s, dmin=None, m=None):
    return torch.from_numpy(O.pack(q_type, q.numpy(), _bits(d), s.numpy(), None if dmin is None else _bits(dmin),
                                   None if m is None else m.numpy()))


def scale_search(x, q_type, rmin=-1.0, rdelta=0.1, nstep=20, **mq):
    _mode(mq)
    d, s, dmin, m = O.scale_search(x.numpy(), q_type, rmin, rdelta, nstep)
    _mode({})
    return _f16(d), torch.from_numpy(s), _f16(dmin), torch.from_numpy(m)


def install(monkeypatch=None):
    """Point the host modules at this backend (they normally call the HIP library)."""
    import sys
    import gptq_gguf_toolkit_amd.gptq as g
    import gptq_gguf_toolkit_amd.quant_utils as qu
    import gptq_gguf_toolkit_amd.quantizer as qz
    import gptq_gguf_toolkit_amd.block_schedule as bs
    import gptq_gguf_toolkit_amd.fast_obq as fo
    me = sys.modules[__name__]
    for mod in (g, qu, qz, bs, fo):
        if monkeypatch is not None:
            monkeypatch.setattr(mod, "_ops", me)
        else:
            mod._ops = me
    for k in list(calls):
        calls[k] = 0
    return me
