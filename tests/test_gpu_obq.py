"""-m gpu: EvoPress' FastOBQ (evopress/src/fast_obq.py) through the C-ABI: gq_obq_h_prepare and
gq_obq_quantize against the reference's own run (G14, bit for bit given the same (W, U)), against the oracle at
Llama shapes, and the FastOBQ handle end to end."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden, triu_unpack

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    from gptq_gguf_toolkit_amd import ops as _ops
    return _ops


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def bits_eq(a, b):
    return np.array_equal(np.ascontiguousarray(a, np.float32).view(np.uint32),
                          np.ascontiguousarray(b, np.float32).view(np.uint32))


@pytest.mark.parametrize("tag", ("g128", "g64sym", "g128b64"))
def test_g14_obq_quantize_bit_exact(ops, tag):
    g = load_golden("g14_fast_obq")
    R, C, gs, sym, block = (int(v) for v in g[f"{tag}_cfg"])
    U = dev(triu_unpack(g[f"{tag}_U_triu"], C))
    for b in (2, 3, 4, 8):
        W = dev(g[f"{tag}_W0"])
        q, sc, ze = ops.obq_quantize(W, U, b, gs, bool(sym), block)
        assert np.array_equal(q.cpu().numpy(), g[f"{tag}_b{b}_q"]), (tag, b)
        assert bits_eq(sc.cpu().numpy(), g[f"{tag}_b{b}_scale"]) and bits_eq(ze.cpu().numpy(), g[f"{tag}_b{b}_zero"])
        grp = np.repeat(np.arange(C // gs), gs)
        want = g[f"{tag}_b{b}_scale"][:, grp] * (g[f"{tag}_b{b}_q"].astype(np.float32) - g[f"{tag}_b{b}_zero"][:, grp])
        assert bits_eq(W.cpu().numpy(), want)  # W becomes the dequantized matrix (fast_obq.py:182)


@pytest.mark.parametrize("tag", ("g64sym", "g128b64"))
def test_g14_obq_h_prepare(ops, tag):
    g = load_golden("g14_fast_obq")
    R, C, *_ = (int(v) for v in g[f"{tag}_cfg"])
    H, W = dev(g[f"{tag}_H_in"]), dev(g[f"{tag}_W_in"])
    U, flag = ops.h_prepare(H, W, 0.01, obq_order=True)
    assert int(flag.item()) == 0 and np.array_equal(W.cpu().numpy(), g[f"{tag}_W0"])
    H2 = H.cpu().numpy()
    assert np.allclose(np.diag(H2), g[f"{tag}_H_after_diag"], rtol=1e-6)
    assert H2[3, 3] == 1.0 and H2[7, 7] == 1.0  # masked AFTER the damping (GPTQ's order gives 1 + damp)
    for r in (3, 7):
        assert np.array_equal(H2[r], g[f"{tag}_H_after_row{r}"]) and np.array_equal(H2[:, r], g[f"{tag}_H_after_row{r}"])
    Uref = triu_unpack(g[f"{tag}_U_triu"], C)
    U = U.cpu().numpy()
    assert np.all(np.tril(U, -1) == 0) and np.abs(U - Uref).max() <= 1e-4 * np.abs(Uref).max()


def _real_u(ops, C, R, seed):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    X = (torch.randn(4096, C, device="cuda", generator=gen) * torch.exp(torch.randn(C, device="cuda", generator=gen) * 0.7))
    H = (2.0 / 4096) * (X.T @ X)
    W = (torch.randn(R, C, device="cuda", generator=gen) * 0.02).half().float().contiguous()
    U, flag = ops.h_prepare(H.contiguous(), W, 0.01, obq_order=True)
    assert int(flag.item()) == 0
    return W, U


@pytest.mark.parametrize("C,gs,sym,block,bits", [
    (4096, 128, False, 128, 4),    # the EvoPress database setting; look-ahead super-blocks
    (4096, 64, True, 128, 3),      # two groups per block: the second grid ignores the block's own feedback
    (4096, 256, False, 128, 2),    # a group spans two blocks
    (4096, 0, False, 128, 4),      # one grid per row
    (2048, 16, False, 64, 8),      # the smallest group; half blocks
    (2048, 128, True, 256, 5),     # a block wider than a segment: block scratch path
    (2048, 2048, False, 128, 4),   # a group wider than a super-block: look-ahead is switched off
    (14336, 128, False, 128, 4),   # down_proj width
])
def test_obq_quantize_llama_shapes_vs_oracle(ops, oracle, C, gs, sym, block, bits):
    """Full-width matrices on the device, a 64-row slice restated by the oracle on the same U: bit for bit."""
    R = 1024 if C <= 4096 else 256
    W, U = _real_u(ops, C, R, seed=C + gs + bits)
    rows = slice(R // 2 - 32, R // 2 + 32)
    W0 = W[rows].cpu().numpy()
    q, sc, ze = ops.obq_quantize(W, U, bits, gs, sym, block)
    Wd, q_ref, s_ref, z_ref = oracle.obq_step(W0, U.cpu().numpy(), bits, gs, sym, block)
    assert np.array_equal(q[rows].cpu().numpy(), q_ref)
    assert bits_eq(sc[rows].cpu().numpy(), s_ref) and bits_eq(ze[rows].cpu().numpy(), z_ref)
    assert bits_eq(W[rows].cpu().numpy(), Wd)
    assert len(np.unique(q_ref)) > min(3, (1 << bits) - 1)


def test_obq_quantize_argument_errors(ops):
    from gptq_gguf_toolkit_amd._cabi import GQError
    W = torch.zeros(64, 256, device="cuda")
    U = torch.eye(256, device="cuda")
    for kw in (dict(bits=9), dict(bits=0), dict(bits=4, group_size=24), dict(bits=4, group_size=96)):
        with pytest.raises(GQError):
            ops.obq_quantize(W, U, **kw)


def test_fast_obq_handle_end_to_end(ops):
    """update() -> quantize() on the device against the reference run: the first group has no Hessian in it (exact);
    U comes from the MFMA chain here and fp32 LAPACK there, so later ints are held to a rate."""
    from make_golden_shim import OBQ_CASES, obq_inputs
    from gptq_gguf_toolkit_amd.fast_obq import FastOBQ
    g = load_golden("g14_fast_obq")
    for tag, R, C, gs, sym, block in OBQ_CASES:
        W, xs = obq_inputs(R, C)
        layer = torch.nn.Linear(C, R, bias=False).cuda()
        layer.weight.data = W.cuda()
        h = FastOBQ(layer, bitwidth_options=[2, 3, 4, 8], group_size=gs, sym=sym, rel_damp=0.01, block_size=block)
        for x in xs:
            h.update(x.cuda())
        q, s, z, perm = h.quantize([2, 3, 4, 8])
        assert perm is None
        for b in (2, 3, 4, 8):
            assert np.array_equal(s[b][:, 0].cpu().numpy(), g[f"{tag}_b{b}_scale"][:, 0])
            assert np.array_equal(q[b][:, 0].cpu().numpy(), g[f"{tag}_b{b}_q"][:, 0])
            mism = float((q[b].cpu().numpy() != g[f"{tag}_b{b}_q"]).mean())
            print(f"[fast_obq {tag} {b} bits] {mism:.4%} ints differ from the reference run")
            assert mism < 0.01
        h.reset()


def test_fast_obq_not_positive_definite_raises(ops):
    from gptq_gguf_toolkit_amd.fast_obq import FastOBQ
    layer = torch.nn.Linear(256, 64, bias=False).cuda()
    h = FastOBQ(layer, bitwidth_options=[4], group_size=128, rel_damp=0.0, block_size=128)
    x = torch.randn(1, 16, 256, device="cuda")  # rank 16 of 256, no damping
    h.update(x)
    with pytest.raises(torch.linalg.LinAlgError):
        h.quantize([4])


def test_obq_quantize_random_small_shapes_vs_oracle(ops, oracle):
    """A seeded sweep over ragged row counts, block sizes (incl. blocks wider than a segment and C itself), group sizes
    (incl. groups wider than a block, and one grid per row) and bit widths: the whole matrix against the oracle."""
    rng = np.random.default_rng(2026)
    for case in range(24):
        C = int(rng.choice([256, 512, 768, 1280]))
        R = int(rng.choice([1, 7, 63, 64, 65, 130, 257]))
        block = int(rng.choice([16, 48, 64, 128, 256, 512, 0]))
        gs = int(rng.choice([g for g in (0, 16, 32, 64, 128, 256) if g == 0 or C % g == 0]))
        sym, bits = bool(rng.integers(2)), int(rng.integers(1, 9))
        X = rng.standard_normal((2 * C, C)).astype(np.float32) * np.exp(rng.standard_normal(C) * 0.5).astype(np.float32)
        H = dev((2.0 / X.shape[0]) * (X.T @ X))
        W = dev((rng.standard_normal((R, C)) * 0.05).astype(np.float32))
        if case % 3 == 0:
            W[:, int(rng.integers(C))] = 0.0
        U, flag = ops.h_prepare(H, W, 0.01, obq_order=True)
        assert int(flag.item()) == 0
        W0, Un = W.cpu().numpy(), U.cpu().numpy()
        q, sc, ze = ops.obq_quantize(W, U, bits, gs, sym, block)
        Wd, q_ref, s_ref, z_ref = oracle.obq_step(W0, Un, bits, gs, sym, block)
        tag = f"case {case}: R={R} C={C} block={block} group={gs} sym={sym} bits={bits}"
        assert np.array_equal(q.cpu().numpy(), q_ref), tag
        assert bits_eq(sc.cpu().numpy(), s_ref) and bits_eq(ze.cpu().numpy(), z_ref), tag
        assert bits_eq(W.cpu().numpy(), Wd), tag
