"""Round 6 GPU tests (VERDICT r05 "next round" items)."""
import os
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from gptq_gguf_toolkit_amd import ops as O
    return O


def _hessian(C, seed, T=None):
    torch.manual_seed(seed)
    T = T or 2 * C
    X = (torch.randn(T, C, device="cuda") * torch.exp(torch.randn(C, device="cuda") * 0.5)).half()
    H = torch.zeros(C, C, device="cuda")
    from gptq_gguf_toolkit_amd import ops as O
    O.h_accumulate(H, X, 0.0, 2.0 / 4)
    return H


# ----------------------------------------------------------------- item 1b: the bottom of gq_h_prepare as resident launches
@pytest.mark.parametrize("C", [256, 384, 896, 1792, 2048, 2176, 4096, 4096 + 896, 14336])
def test_resident_sub_problems_compute_the_same_u(ops, C):
    """VERDICT r05 next #1b: the sub-problems of the Cholesky recursion below the image levels can run as ONE resident launch
    each (option chol_sub; csrc/gq_cholsub.hpp: a task graph of the launches the recursion would make, same tile functions;
    measured neutral inside a step and slower alone, hence off by default -- DESIGN.md 0a).  U must not
    change by a bit against the launch-by-launch recursion (option chol_sub = 0), whatever the number of workgroups that
    execute the graph -- one workgroup walks it alone (the executor never waits for a workgroup that is not running)."""
    H = _hessian(C, C)
    W = torch.randn(64, C, device="cuda")
    with ops.options(chol_sub=0):
        U0, f0 = ops.h_prepare(H.clone(), W.clone(), 0.01)
    assert int(f0.item()) == 0 and bool(torch.isfinite(U0).all())
    for wgs in ([48, 1, 5, 256] if C <= 4096 else [48, 16]):
        with ops.options(chol_sub=16, chol_sub_wgs=wgs):
            U1, f1 = ops.h_prepare(H.clone(), W.clone(), 0.01)
        assert int(f1.item()) == 0
        assert torch.equal(U0, U1), f"C={C} wgs={wgs}: {(U0 != U1).float().mean().item():.3%} of U differs"
    with ops.options(chol_sub=4):  # narrower roots: more launches, the same products
        U2, _ = ops.h_prepare(H.clone(), W.clone(), 0.01)
    assert torch.equal(U0, U2)


def test_resident_sub_problems_never_read_unwritten_scratch_and_flag_singular(ops):
    """The executor inherits the recursion's contracts: scratch poisoned with NaN patterns gives the same U; a Hessian that
    is not positive definite raises the flag and U = I (gptq.py:321-323)."""
    C = 2048 + 896
    H = _hessian(C, 11)
    W = torch.randn(64, C, device="cuda")
    with ops.options(chol_sub=16):
        U0, f0 = ops.h_prepare(H.clone(), W.clone(), 0.01)
    with ops.options(chol_sub=16, chol_poison=1):
        U1, f1 = ops.h_prepare(H.clone(), W.clone(), 0.01)
    assert int(f0.item()) == 0 and int(f1.item()) == 0 and torch.equal(U0, U1)
    Hs = H.clone()
    Hs[5, 5] = -1.0
    with ops.options(chol_sub=16):
        Ui, fi = ops.h_prepare(Hs, W.clone(), 0.0)
    assert int(fi.item()) == 1 and torch.equal(Ui, torch.eye(C, device="cuda"))


def test_resident_sub_problems_on_concurrent_streams(ops):
    """Four chains at a time (a block's schedule): resident launches of different Hessians next to each other on four streams
    -- more claimed workgroups than some launches will find CUs for at once -- give each the U it gets alone."""
    Cs = [1792, 2048, 1024, 3584]
    Hs = [_hessian(C, 20 + i) for i, C in enumerate(Cs)]
    Ws = [torch.randn(64, C, device="cuda") for C in Cs]
    alone = [ops.h_prepare(H.clone(), W.clone(), 0.01)[0] for H, W in zip(Hs, Ws)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in Cs]
    outs = [None] * len(Cs)
    for rep in range(3):
        with ops.options(chol_sub=16, chol_sub_wgs=96):
            for i, s in enumerate(streams):
                with torch.cuda.stream(s):
                    outs[i] = ops.h_prepare(Hs[i].clone(), Ws[i].clone(), 0.01)[0]
        torch.cuda.synchronize()
        for a, b in zip(alone, outs):
            assert torch.equal(a, b)
