#!/usr/bin/env python3
"""Generate golden vectors by RUNNING THE REFERENCE (read-only at /root/reference).

Runs only in the build container (the reference never travels to the GPU box).
Nothing from the reference is copied: this script imports
quant/gptq/src/{gptq,quant_utils,packing_utils,quantizer}.py, feeds them seeded
inputs and stores inputs + outputs as small .npz fixtures next to this file.

Two environment shims, both recorded in every fixture's `meta`:

1. `gguf` stub.  The reference imports gguf-py 0.17.1 (not installed, not
   vendored) for three constants only: QK_K = 256, the GGMLQuantizationType ids
   10..14 and GGML_QUANT_SIZES[Qn_K] = (256, {84,110,144,176,210}).

2. IEEE sqrt.  torch.sqrt on CPU goes through MKL VML `vsSqrt` (VML_HA), which is
   NOT correctly rounded: 0.6 % of results are 1 ulp below sqrtf (measured here,
   deterministic per value).  The reference's CUDA path, the oracle and the HIP
   kernels all use the IEEE-754 correctly rounded sqrt.  The "ieee" fixtures are
   produced with `src.quant_utils.torch.sqrt` routed through numpy's correctly
   rounded sqrt (module-local proxy; the reference source is untouched); the
   "mkl" fixtures are produced with stock torch and are used by the tests only
   to REPORT the resulting flip rate (make_k_quants near-ties), never as the
   bit-exact anchor.

Usage:  python tests/golden/make_golden.py   (rewrites tests/golden/*.npz)
"""
import enum
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/quant/gptq"


def _install_gguf_stub():
    g = types.ModuleType("gguf")
    c = types.ModuleType("gguf.constants")

    class GGMLQuantizationType(enum.IntEnum):
        Q2_K = 10
        Q3_K = 11
        Q4_K = 12
        Q5_K = 13
        Q6_K = 14

    c.QK_K = 256
    c.GGMLQuantizationType = GGMLQuantizationType
    c.GGML_QUANT_SIZES = {
        GGMLQuantizationType.Q2_K: (256, 84), GGMLQuantizationType.Q3_K: (256, 110),
        GGMLQuantizationType.Q4_K: (256, 144), GGMLQuantizationType.Q5_K: (256, 176),
        GGMLQuantizationType.Q6_K: (256, 210)}
    g.constants = c
    g.GGMLQuantizationType = GGMLQuantizationType
    sys.modules["gguf"] = g
    sys.modules["gguf.constants"] = c
    sys.path.insert(0, REF)


_install_gguf_stub()
import src.quant_utils as ref_qu  # noqa: E402
import src.packing_utils as ref_pk  # noqa: E402
from src.gptq import GPTQ as RefGPTQ  # noqa: E402
from src.quantizer import Quantizer as RefDriver  # noqa: E402

T = ref_qu.GGMLQuantizationType


class _TorchIEEE:
    """torch proxy: sqrt is correctly rounded (numpy), everything else is torch."""

    def __getattr__(self, k):
        return getattr(torch, k)

    @staticmethod
    def sqrt(t):
        if t.dtype not in (torch.float32, torch.float64):
            return torch.sqrt(t)  # reduced floats never reach MKL VML (Vectorized<float>::sqrt is IEEE)
        return torch.from_numpy(np.sqrt(t.detach().cpu().numpy())).to(t.device)


def set_sqrt(mode):
    ref_qu.torch = _TorchIEEE() if mode == "ieee" else torch


META = dict(torch=torch.__version__, cpu_capability=torch.backends.cpu.get_cpu_capability(),
            reference="IST-DASLab/gptq-gguf-toolkit @ 2025-09-19", numpy=np.__version__)


def save(name, **arrs):
    arrs["meta"] = np.array(repr(META))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrs)
    sz = os.path.getsize(os.path.join(HERE, name + ".npz"))
    print(f"  wrote {name}.npz ({sz / 1024:.0f} KiB)")


def u16(t):
    return t.contiguous().view(torch.int16).numpy().view(np.uint16)


def configured(qt, **kw):
    bits, _, smq, G, SG, sdt, _ = ref_qu.GGML_QUANT_SIZES[qt]
    q = ref_qu.Quantizer()
    q.configure(bits=bits, scale_maxq=smq, super_group_size=SG, group_size=G, group_type=sdt, **kw)
    return q, bits, G


def edge_rows(x, G):
    """rows 0..9 of x[n,G] get the edge cases of SURVEY 8c/G1."""
    x[0] = 0.01                       # all equal, positive
    x[1] = 0.0                        # all zero
    x[2] = x[2].abs()                 # all positive (min clamps to 0)
    x[3, 5] = 3.0                     # one outlier
    x[4] = 1e-41                      # fp32 denormals
    x[5] = -x[5].abs()                # all negative
    x[6] = 1e-7 * torch.randn(G)      # tiny magnitudes (D <= eps region)
    x[7] = -0.02                      # all equal, negative
    x[8, : G // 2] = 0.0              # half zeros
    x[9] = torch.linspace(-0.05, 0.05, G)
    return x


def g1_make_quants():
    out = {}
    for mode in ("ieee", "mkl"):
        set_sqrt(mode)
        for qt in T:
            torch.manual_seed(100 + int(qt))
            q, bits, G = configured(qt)
            x = edge_rows(torch.randn(512, G) * 0.02, G)
            fn = q.make_k_quants if qt in (T.Q2_K, T.Q4_K, T.Q5_K) else q.make_quants
            sc, ze = fn(x.clone())
            if mode == "ieee":
                out[f"{qt.name}_x"] = x.numpy()
            out[f"{qt.name}_{mode}_scale"] = sc.flatten().numpy()
            out[f"{qt.name}_{mode}_zero"] = ze.flatten().numpy()
        # non-default search parameters (ieee only)
    set_sqrt("ieee")
    torch.manual_seed(7)
    q, bits, G = configured(T.Q4_K, rmin=-0.5, rdelta=0.05, nstep=10)
    x = torch.randn(256, G) * 0.05
    sc, ze = q.make_k_quants(x.clone())
    out["Q4_K_alt_x"], out["Q4_K_alt_scale"], out["Q4_K_alt_zero"] = x.numpy(), sc.numpy(), ze.numpy()
    q, bits, G = configured(T.Q4_K, nstep=0)
    sc, ze = q.make_k_quants(x.clone())
    out["Q4_K_nstep0_scale"], out["Q4_K_nstep0_zero"] = sc.numpy(), ze.numpy()
    save("g1_make_quants", **out)


def g2_scale_search():
    out = {}
    for mode in ("ieee", "mkl"):
        set_sqrt(mode)
        for qt in T:
            torch.manual_seed(200 + int(qt))
            q, bits, G = configured(qt)
            x = torch.randn(64, 256) * 0.02
            x[0] = 0.0
            x[1] = x[1].abs()           # all-positive super-group: dmin = -0.0 candidates
            x[2, :64] *= 40.0           # one loud group
            x[3] = 0.5
            d, s, dmin, m = q.get_scale_and_zero(x.clone(), qt)
            if mode == "ieee":
                out[f"{qt.name}_x"] = x.numpy()
            out[f"{qt.name}_{mode}_d"], out[f"{qt.name}_{mode}_dmin"] = u16(d), u16(dmin)
            out[f"{qt.name}_{mode}_s"], out[f"{qt.name}_{mode}_m"] = s.numpy(), m.numpy()
    save("g2_scale_search", **out)


def g3_elementwise():
    set_sqrt("ieee")
    out = {}
    torch.manual_seed(3)
    n = 4096
    for qt in T:
        _, clamp, smq, G, _, sdt, _ = ref_qu.GGML_QUANT_SIZES[qt]
        d = (torch.rand(n) * 2e-3).half()
        dmin = (torch.rand(n) * 2e-3).half() if qt in (T.Q2_K, T.Q4_K, T.Q5_K) else torch.zeros(n).half()
        s = torch.randint(0, smq + 1, (n,)).to(sdt)
        m = torch.randint(0, smq + 1, (n,)).to(sdt) if qt in (T.Q2_K, T.Q4_K, T.Q5_K) else torch.zeros(n).to(sdt)
        s[:64] = 0                                    # d*s == 0 -> eps clamp
        x = torch.randn(n) * 0.05
        # exact .5 ties: x = (k + 0.5) * d*s - dmin*m  (where representable)
        ds = d.float() * s
        x[64:512] = (torch.randint(-3, 8, (448,)).float() + 0.5) * ds[64:512] - dmin.float()[64:512] * m[64:512]
        q = ref_qu.quantize(x, d, s, dmin, m, clamp)
        w = ref_qu.dequantize(q, d, s, dmin, m)
        out[f"{qt.name}_x"], out[f"{qt.name}_d"], out[f"{qt.name}_dmin"] = x.numpy(), u16(d), u16(dmin)
        out[f"{qt.name}_s"], out[f"{qt.name}_m"] = s.numpy(), m.numpy()
        out[f"{qt.name}_q"], out[f"{qt.name}_w"] = q.numpy(), w.numpy()
    save("g3_elementwise", **out)


def _mk_layer(R, C, seed):
    torch.manual_seed(seed)
    layer = nn.Linear(C, R, bias=False)
    layer.weight.data = (torch.randn(R, C) * 0.02).half().float()
    return layer


def _calib(C, n_batches, L, seed, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    sig = torch.exp(torch.randn(C, generator=g) * 0.5)
    sig[:: max(C // 4, 1)] *= 20.0  # a few loud channels
    return [(torch.randn(1, L, C, generator=g) * sig).to(dtype) for _ in range(n_batches)]


def _triu_pack(U):
    iu = np.triu_indices(U.shape[0])
    assert np.all(np.tril(U, -1) == 0)
    return U[iu].astype(np.float32)


def g4_g5_hessian():
    set_sqrt("ieee")
    out = {}
    C, R = 256, 32
    layer = _mk_layer(R, C, 40)
    for dt, tag in ((torch.float32, "f32"), (torch.float16, "f16"), (torch.bfloat16, "bf16")):
        g = RefGPTQ(layer, rel_damp=0.01, block_size=128)
        xs = _calib(C, 3, 64, 41, dt)
        for x in xs:
            g.update(x)
        out[f"X_{tag}"] = np.stack([x[0].float().numpy() for x in xs])  # values exactly representable
        out[f"H_{tag}"] = g.H.numpy().copy()
    # 2-D input (MoE expert style): num_samples counts tokens (gptq.py:88)
    g = RefGPTQ(layer)
    x2 = _calib(C, 2, 48, 42)
    for x in x2:
        g.update(x[0])
    out["X_2d"] = np.stack([x[0].numpy() for x in x2])
    out["H_2d"] = g.H.numpy().copy()
    out["n_2d"] = np.array(g.num_samples)

    # G5: pre_step + _prepare, with a dead input channel and an all-zero weight column
    layer = _mk_layer(R, C, 43)
    layer.weight.data[:, 17] = 0.0
    g = RefGPTQ(layer, rel_damp=0.01, block_size=128)
    xs = _calib(C, 3, 256, 44)
    for x in xs:
        x[..., 5] = 0.0  # dead channel: H[5,5] == 0
        g.update(x)
    out["prep_H_in"] = g.H.numpy().copy()
    out["prep_W_in"] = layer.weight.data.numpy().copy()
    g.quantization_pre_step()
    out["prep_W_after_prestep"] = g.W.numpy().copy()
    U = g._prepare()
    out["prep_U_triu"] = _triu_pack(U.numpy())
    out["prep_H_after_diag"] = np.diag(g.H.numpy()).copy()
    out["prep_H_after_row5"] = g.H.numpy()[5].copy()
    out["prep_H_after_row17"] = g.H.numpy()[17].copy()
    # singular H -> identity fallback (rel_damp = 0, rank-deficient H)
    layer = _mk_layer(8, 256, 45)
    g = RefGPTQ(layer, rel_damp=0.0)
    xs = torch.randn(1, 16, 256)
    g.update(xs)  # rank 16 < 256
    out["sing_X"] = xs[0].numpy()
    g.quantization_pre_step()
    U = g._prepare()
    out["sing_U_is_identity"] = np.array(bool(torch.equal(U, torch.eye(256))))
    out["sing_flag"] = np.array(bool(getattr(g, "issue_non_invertible", False)))
    save("g4_g5_hessian", **out)


def _run_step(layer, xs, qt, block, static, mode):
    set_sqrt(mode)
    g = RefGPTQ(layer, rel_damp=0.01, block_size=block, static_groups=static)
    for x in xs:
        g.update(x)
    g.quantization_pre_step()
    W0 = g.W.numpy().copy()
    cap = {}
    orig = g._prepare

    def prep():
        u = orig()
        cap["U"] = u.clone()
        return u

    g._prepare = prep
    q, d, s, dmin, m = g.step(qt)
    return W0, cap["U"].numpy(), (q.numpy(), u16(d), s.numpy(), u16(dmin), m.numpy(), g.W.numpy().copy())


def _pack_ref(qt, q, d, s, dmin, m):
    a = [torch.from_numpy(np.array(v)) for v in (q, d, s, dmin, m)]
    a[1] = a[1].view(torch.float16) if a[1].dtype != torch.float16 else a[1]
    a[3] = a[3].view(torch.float16) if a[3].dtype != torch.float16 else a[3]
    if qt == T.Q2_K:
        return ref_pk.pack_Q2K(*a)
    if qt == T.Q3_K:
        return ref_pk.pack_Q3K(a[0], a[1], a[2])
    if qt == T.Q4_K:
        return ref_pk.pack_Q4K(*a)
    if qt == T.Q5_K:
        return ref_pk.pack_Q5K(*a)
    return ref_pk.pack_Q6K(a[0], a[1], a[2])


def g6_g7_step_and_pack():
    out = {}
    # case A: 64 x 512, every type, block 128 (+ variants), one shared (W, U)
    R, C = 64, 512
    layer = _mk_layer(R, C, 60)
    xs = _calib(C, 4, 256, 61)
    cases = [(qt, 128, False) for qt in T] + [(T.Q4_K, 64, False), (T.Q4_K, 128, True), (T.Q2_K, 128, True),
                                              (T.Q5_K, None, False), (T.Q6_K, 256, False)]
    for qt, block, static in cases:
        tag = f"A_{qt.name}_b{block}_s{int(static)}"
        W0, U, res = _run_step(layer, xs, qt, block, static, "ieee")
        if "A_W0" not in out:
            out["A_W0"], out["A_U_triu"] = W0, _triu_pack(U)
        else:
            assert np.array_equal(out["A_W0"], W0) and np.array_equal(out["A_U_triu"], _triu_pack(U))
        for k, v in zip(("q", "d", "s", "dmin", "m", "Wdeq"), res):
            if k == "Wdeq":
                # Wdeq is exactly dequantize(q,...) (gptq.py:266) -- store a checksum only
                out[f"{tag}_Wdeq_sum"] = np.array(v.astype(np.float64).sum())
                out[f"{tag}_Wdeq_head"] = v[:4, :64].copy()
            else:
                out[f"{tag}_{k}"] = v
        if block == 128 and not static:
            out[f"{tag}_packed"] = _pack_ref(qt, *res[:5])  # G7
            # flip rate vs stock-MKL sqrt, for the record
            _, _, res_mkl = _run_step(layer, xs, qt, block, static, "mkl")
            out[f"{tag}_mklsqrt_q"] = res_mkl[0]
            out[f"{tag}_mklsqrt_s"] = res_mkl[2]
    # case B: 96 x 768 Q4_K / Q3_K block 128
    R, C = 96, 768
    layer = _mk_layer(R, C, 62)
    xs = _calib(C, 4, 384, 63)
    for qt in (T.Q4_K, T.Q3_K):
        tag = f"B_{qt.name}_b128_s0"
        W0, U, res = _run_step(layer, xs, qt, 128, False, "ieee")
        if "B_W0" not in out:
            out["B_W0"], out["B_U_triu"] = W0, _triu_pack(U)
        for k, v in zip(("q", "d", "s", "dmin", "m"), res[:5]):
            out[f"{tag}_{k}"] = v
        out[f"{tag}_packed"] = _pack_ref(qt, *res[:5])
    save("g6_g7_step_and_pack", **out)


def g8_g9_rtn_dequant():
    set_sqrt("ieee")
    out = {}
    R, C = 48, 768
    torch.manual_seed(90)
    W = torch.randn(R, C) * 0.02
    W[:, 5] = W[:, 5].abs() * 30
    W[3, :256] = W[3, :256].abs()
    W[4] = 0.0
    out["W"] = W.numpy()
    drv = RefDriver.__new__(RefDriver)
    drv.quantizer_kwargs = {}
    for qt in T:
        q, d, s, dmin, m = drv._quant_non_block_module(W.clone(), qt)
        deq = ref_qu.dequantize_linear_weight(qt, q, d, s, dmin, m)
        out[f"{qt.name}_q"], out[f"{qt.name}_d"], out[f"{qt.name}_dmin"] = q.numpy(), u16(d), u16(dmin)
        out[f"{qt.name}_s"], out[f"{qt.name}_m"] = s.numpy(), m.numpy()
        out[f"{qt.name}_deq"] = deq.numpy()
        out[f"{qt.name}_packed"] = _pack_ref(qt, q.numpy(), u16(d), s.numpy(), u16(dmin), m.numpy())
    # model-dtype RTN (quantizer.py:109,195 pass module.weight un-cast): bf16 / fp16
    for dt, tag in ((torch.float16, "f16"), (torch.bfloat16, "bf16")):
        Wl = W.to(dt)
        out[f"W_{tag}"] = Wl.float().numpy()
        for qt in T:
            q, d, s, dmin, m = drv._quant_non_block_module(Wl.clone(), qt)
            out[f"{tag}_{qt.name}_q"] = q.numpy()
            out[f"{tag}_{qt.name}_d"], out[f"{tag}_{qt.name}_dmin"] = u16(d), u16(dmin)
            out[f"{tag}_{qt.name}_s"], out[f"{tag}_{qt.name}_m"] = s.numpy(), m.numpy()
    save("g8_g9_rtn_dequant", **out)


def g11_act_order():
    """GPTQ.step with act_order=True (implies static_groups, gptq.py:45-46): inputs (W after pre_step, H after
    pre_step), the permutation and the permuted U the reference used, outputs in ORIGINAL column order."""
    out = {}
    R, C = 64, 512
    layer = _mk_layer(R, C, 110)
    xs = _calib(C, 4, 256, 111)
    for qt, block in ((T.Q4_K, 128), (T.Q2_K, 128), (T.Q6_K, 64), (T.Q5_K, 128)):
        set_sqrt("ieee")
        g = RefGPTQ(layer, rel_damp=0.01, block_size=block, static_groups=True, act_order=True)
        for x in xs:
            g.update(x)
        g.quantization_pre_step()
        W0, H0 = g.W.numpy().copy(), g.H.numpy().copy()
        cap = {}
        orig = g._prepare

        def prep():
            cap["Wp"] = g.W.numpy().copy()   # permuted W entering _prepare
            u = orig()
            cap["U"] = u.clone()
            return u

        g._prepare = prep
        q, d, s, dmin, m = g.step(qt)
        perm = np.argsort(-np.diag(H0), kind="stable").astype(np.int32)
        assert np.array_equal(cap["Wp"], W0[:, perm]), "reference perm != stable descending argsort of diag(H)"
        tag = f"{qt.name}_b{block}"
        if "W0" not in out:
            out["W0"], out["H0"], out["perm"] = W0, H0, perm
        if "U_triu" not in out:  # U depends on (H[perm][:, perm], zero columns of W) only
            out["U_triu"] = _triu_pack(cap["U"].numpy())
        else:
            assert np.array_equal(out["U_triu"], _triu_pack(cap["U"].numpy()))
        for k, v in zip(("q", "d", "s", "dmin", "m"), (q.numpy(), u16(d), s.numpy(), u16(dmin), m.numpy())):
            out[f"{tag}_{k}"] = v
    save("g11_act_order", **out)


# ----------------------------------------------------------------------------- G13 two ranks (gloo)
def _g13_worker(rank, world, port, ret):
    """One rank of the REFERENCE's calibration-sharded run: own shard -> update, quantization_pre_step
    (all_reduce AVG, gptq.py:131-132), step (rank 0 computes, broadcasts, gptq.py:158,286-293)."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    set_sqrt("ieee")
    R, C = 32, 256
    out = {}
    for tag, qt in (("Q4_K", T.Q4_K), ("Q6_K", T.Q6_K)):
        layer = _mk_layer(R, C, 130)
        xs = _calib(C, 4, 64, 131)
        xs[0][..., 9] = 0.0
        xs[1][..., 9] = 0.0
        xs[2][..., 9] = 0.0
        xs[3][..., 9] = 0.0  # a dead channel on every rank
        g = RefGPTQ(layer, rel_damp=0.01, block_size=128)
        for x in xs[rank * 2:(rank + 1) * 2]:  # contiguous shard (quant.py:177-179)
            g.update(x)
        out["H_local"] = g.H.numpy().copy()
        g.quantization_pre_step()
        out["H_reduced"] = g.H.numpy().copy()
        cap = {}
        orig = g._prepare

        def prep(orig=orig, cap=cap):
            u = orig()
            cap["U"] = u.clone()
            return u

        g._prepare = prep
        q, d, s, dmin, m = g.step(qt)
        if rank == 0:
            out[f"{tag}_U_triu"] = _triu_pack(cap["U"].numpy())
        for k, v in zip(("q", "d", "s", "dmin", "m"), (q.numpy(), u16(d), s.numpy(), u16(dmin), m.numpy())):
            out[f"{tag}_{k}"] = v.copy()
        if tag == "Q4_K":
            out["W0"] = layer.weight.data.numpy().copy()
            out["X"] = np.stack([x[0].numpy() for x in xs])
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


def g13_two_rank():
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_g13_worker, args=(2, 29871, ret), nprocs=2, join=True)
    r0, r1 = ret[0], ret[1]
    out = {"W0": r0["W0"], "X": r0["X"], "H_local_rank0": r0["H_local"], "H_local_rank1": r1["H_local"],
           "H_reduced": r0["H_reduced"]}
    assert np.array_equal(r0["H_reduced"], r1["H_reduced"])
    for tag in ("Q4_K", "Q6_K"):
        out[f"{tag}_U_triu"] = r0[f"{tag}_U_triu"]
        for k in ("q", "d", "s", "dmin", "m"):
            assert np.array_equal(r0[f"{tag}_{k}"], r1[f"{tag}_{k}"]), "ranks disagree in the reference run"
            out[f"{tag}_{k}"] = r0[f"{tag}_{k}"]
    save("g13_two_rank", **out)


def g15_rtn_mse():
    """_quant_non_block_module with quantizer_kwargs["quant_scale"] = "mse" (quantizer.py:293-295 forwards it; the
    grid / maxshrink stay at Quantizer.configure's defaults 100 / 0.8): embed-like weights in fp32 / fp16 / bf16 with
    entries beyond +-32 present -- below that every candidate of the grid search quantizes to q_int = 0 and the branch
    returns the absmax scale (g12) -- for the two types that use make_quants, plus Q4_K (make_k_quants ignores it).
    Found on the way: with fp16 / bf16 weights the reference RAISES for Q3_K / Q6_K (recorded as `<tag>_<type>_raises`)."""
    set_sqrt("ieee")
    out = {}
    R, C = 40, 512
    torch.manual_seed(150)
    W = torch.randn(R, C) * 0.05
    W[:, 7] *= 900.0             # one hot column: |w| up to ~100
    W[5, 256:272] = torch.linspace(-60.0, 75.0, 16)
    W[6, :16] = 33.0             # constant group beyond 32
    W[7] = 0.0
    W[8, 100] = -48.5
    drv = RefDriver.__new__(RefDriver)
    for dt, tag in ((torch.float32, "f32"), (torch.float16, "f16"), (torch.bfloat16, "bf16")):
        Wl = W.to(dt)
        out[f"W_{tag}"] = Wl.float().numpy()
        for qt in (T.Q3_K, T.Q6_K, T.Q4_K):
            drv.quantizer_kwargs = {"quant_scale": "mse"}
            try:
                q, d, s, dmin, m = drv._quant_non_block_module(Wl.clone(), qt)
            except RuntimeError as e:
                # fp16 / bf16 weights + make_quants: quant_utils.py:165 creates min_loss in fp32, :187 index_puts the
                # model-dtype loss into it -> "Index put requires the source and destination dtypes match"
                out[f"{tag}_{qt.name}_raises"] = np.array(str(e).splitlines()[0])
                continue
            out[f"{tag}_{qt.name}_q"] = q.numpy()
            out[f"{tag}_{qt.name}_d"], out[f"{tag}_{qt.name}_dmin"] = u16(d), u16(dmin)
            out[f"{tag}_{qt.name}_s"], out[f"{tag}_{qt.name}_m"] = s.numpy(), m.numpy()
            drv.quantizer_kwargs = {}
            qa, da, sa, _, _ = drv._quant_non_block_module(Wl.clone(), qt)
            # how much the branch matters on this weight (reported by the tests, not asserted)
            out[f"{tag}_{qt.name}_differs_from_absmax"] = np.array(float((qa != q).float().mean()))
    save("g15_rtn_mse", **out)


def g16_conv_handle():
    """The _ConvNd branch of the handle (gptq.py:76, 96-104, 138-139): a small nn.Conv2d through GPTQ.update (nn.Unfold
    patches as rows) and GPTQ.quantize; d_col = C_in * kh * kw = 256.  Stored: the conv weight, the three input
    batches, H after the updates, W / H entering step, U, and the 5-tuple for two types."""
    set_sqrt("ieee")
    out = {}
    torch.manual_seed(160)
    conv = nn.Conv2d(64, 48, kernel_size=2, stride=2, padding=1, bias=False)
    conv.weight.data = (torch.randn_like(conv.weight) * 0.05).half().float()
    out["weight"] = conv.weight.detach().numpy().copy()
    out["conv"] = np.array([64, 48, 2, 2, 1])  # in, out, kernel, stride, padding
    xs = [(torch.randn(2, 64, 9, 9) * torch.exp(torch.randn(1, 64, 1, 1) * 0.4)).half().float() for _ in range(3)]
    out["x"] = np.stack([x.numpy() for x in xs])
    for qt in (T.Q4_K, T.Q6_K):
        g = RefGPTQ(conv, rel_damp=0.01, block_size=128)
        for x in xs:
            g.update(x)
        out["H_updated"] = g.H.numpy().copy()
        out["num_samples"] = np.array(g.num_samples)
        g.quantization_pre_step()
        out["W0"], out["H0"] = g.W.numpy().copy(), g.H.numpy().copy()
        cap = {}
        orig = g._prepare

        def prep():
            u = orig()
            cap["U"] = u.clone()
            return u

        g._prepare = prep
        q, d, s, dmin, m = g.step(qt)
        out["U_triu"] = _triu_pack(cap["U"].numpy())
        for k, v in zip(("q", "d", "s", "dmin", "m"), (q.numpy(), u16(d), s.numpy(), u16(dmin), m.numpy())):
            out[f"{qt.name}_{k}"] = v
    save("g16_conv_handle", **out)


def _main_all():
    torch.set_num_threads(8)
    for fn in (g1_make_quants, g2_scale_search, g3_elementwise, g4_g5_hessian, g6_g7_step_and_pack,
               g8_g9_rtn_dequant, g11_act_order, g12_mse_scale, g13_two_rank, g15_rtn_mse, g16_conv_handle):
        print(fn.__name__)
        fn()
    g10_driver()


# ----------------------------------------------------------------------------- G10 driver tree
def tiny_llama(seed=0, dtype=torch.float32):
    """Seeded random LlamaForCausalLM used by the driver fixtures AND by tests/test_driver_gpu.py."""
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, vocab_size=512, max_position_embeddings=128, rms_norm_eps=1e-5,
                      tie_word_embeddings=False, attn_implementation="eager")
    torch.manual_seed(seed)
    model = LlamaForCausalLM(cfg).to(dtype)
    model.eval()
    return model


def tiny_calib(n=8, L=64, vocab=512, seed=1):
    g = torch.Generator().manual_seed(seed)
    return [torch.randint(0, vocab, (1, L), generator=g) for _ in range(n)]


MIXED = {"q_proj": "Q3_K", "k_proj": "Q2_K", "v_proj": "Q4_K", "o_proj": "Q5_K", "gate_proj": "Q6_K",
         "down_proj": "Q3_K", "up_proj": "Q4_K", "embed_tokens": "Q6_K", "lm_head": "Q6_K"}  # README.md:94-106


def g12_mse_scale():
    """make_quants with quant_scale="mse" (quant_utils.py:164-191).  The grid search divides by
    scale1.clamp_min(1e-9).round() (:180): for every group whose absmax scale is <= 0.5 that divisor is 0, all 81
    candidates quantize to q_int = 0 with the same loss, and the first one (alpha = 1, the absmax scale) is kept --
    the branch returns exactly the absmax result.  Recorded: weight-like panels (N(0, 0.02^2), N(0, 0.3^2), edge
    rows) for Q3_K / Q6_K, outputs of get_scale_and_zero in both modes, plus one wide panel (sigma = 30) where
    the branch really differs."""
    out = {}
    set_sqrt("ieee")
    for qt in (T.Q3_K, T.Q6_K):
        for tag, sigma in (("w", 0.02), ("mid", 0.3), ("wide", 30.0)):
            torch.manual_seed(1200 + int(qt))
            x = torch.randn(64, 256) * sigma
            if tag == "w":
                x[:10] = edge_rows(x[:10].reshape(-1, 16).clone(), 16)[:160].reshape(10, 256)
            out[f"{qt.name}_{tag}_x"] = x.numpy()
            for mode in ("absmax", "mse"):
                q, bits, G = configured(qt, quant_scale=ref_qu.QuantizationScale(mode))
                d, s, dmin, m = q.get_scale_and_zero(x.clone(), qt)
                out[f"{qt.name}_{tag}_{mode}_d"] = d.view(torch.int16).numpy().view(np.uint16)
                out[f"{qt.name}_{tag}_{mode}_s"] = s.numpy()
    save("g12_mse_scale", **out)


def g10_driver():
    import tempfile
    set_sqrt("ieee")
    model = tiny_llama()
    data = [([], {"input_ids": ids}) for ids in tiny_calib()]
    qc = {k: T[v] for k, v in MIXED.items()}
    out = {}
    with tempfile.TemporaryDirectory() as td:
        drv = RefDriver(model, data_loader=data, quantizable_modules=r".*layers.*((q|k|v|o|gate|up|down)_proj)$",
                        quantizer_kwargs=dict(rel_damp=0.01, block_size=128, act_order=False, quant_scale="absmax",
                                              static_groups=False, rmin=-1.0, rdelta=0.1, nstep=20, verbose=False),
                        pre_block_modules=["model.embed_tokens"], block_modules="model.layers",
                        post_block_modules=["lm_head"], quant_non_block_modules=True, device="cpu", save_dir=td)
        drv.quantize(qc)
        names = sorted(os.listdir(td))
        out["names"] = np.array(names)
        for n in names:
            d = torch.load(os.path.join(td, n, "data.pth"), weights_only=True)
            out[f"{n}|q_type"] = np.array(d["q_type"])
            out[f"{n}|qweight"] = d["qweight"].numpy()
            out[f"{n}|d"], out[f"{n}|dmin"] = u16(d["super_group_scale"]), u16(d["super_group_zero"])
            out[f"{n}|s"], out[f"{n}|m"] = d["group_scale_quant"].numpy(), d["group_zero_quant"].numpy()
    # logits of the quantized model on the first calibration sample (end-to-end sanity anchor)
    with torch.no_grad():
        out["logits_head"] = model(tiny_calib()[0]).logits[0, :4, :16].numpy()
    save("g10_driver", **out)


if __name__ == "__main__":
    if "g10" in sys.argv[1:]:
        g10_driver()
    elif "g11" in sys.argv[1:]:
        g11_act_order()
    elif "g12" in sys.argv[1:]:
        g12_mse_scale()
    elif "g13" in sys.argv[1:]:
        g13_two_rank()
    elif "g15" in sys.argv[1:]:
        g15_rtn_mse()
    elif "g16" in sys.argv[1:]:
        g16_conv_handle()
    else:
        _main_all()
