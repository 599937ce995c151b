"""Correlated / low-rank calibration activations for the tolerance-class tests of K3 (gptq.py:304-324,
linalg_utils.py:8-12) -- VERDICT r03 weak #1.  iid Gaussian activations give a Hessian that is nearly the identity once
it is equilibrated; real activations are strongly correlated, so after the 1 % damping the matrix the chain factorises
has a condition number of 1e5-1e7.  `X = Z_r A + eps Z` with rank r << C, plus a few "massive activation" channels.
Test infrastructure only (GPU tests, bench.py's checker leg, profiles/)."""
import math

import torch


def correlated_x(T, C, rank, eps, seed, massive=6, massive_gain=1e3, mean_shift=0.0, dtype=torch.float16, device="cuda"):
    """massive: channels scaled by massive_gain (they dominate mean(diag H), hence the damping: the damped matrix is then
    WELL conditioned).  mean_shift: a constant offset m * v (v ~ N(0, 1) per channel) added to every token -- real
    activations are not centred -- which gives the equilibrated Hessian one eigenvalue of order C: with massive = 0 the
    damped matrix has cond ~ C / rel_damp = 1e5 - 1e6."""
    g = torch.Generator(device=device).manual_seed(seed)
    A = torch.randn(rank, C, device=device, generator=g) / math.sqrt(rank)
    X = torch.randn(T, rank, device=device, generator=g) @ A
    del A
    X.add_(torch.randn(T, C, device=device, generator=g), alpha=eps)
    if mean_shift:
        X.add_(torch.randn(1, C, device=device, generator=g), alpha=mean_shift)
    if massive:
        idx = torch.randperm(C, device=device, generator=g)[:massive]
        X[:, idx] *= massive_gain
    # fp16 holds |x| <= 65504: the massive channels stay below 1e3 * 6 sigma
    return X.to(dtype)


def env(**kv):
    """Library options for the block (gptq_gguf_toolkit_amd._cabi.options)."""
    from gptq_gguf_toolkit_amd import ops
    return ops.options(**kv)


# the three forms of the level-3 work of the chain, as library options (`with env(**CHAIN_MODES[mode]):`)
CHAIN_MODES = {
    "default": {},                                   # image GEMMs (row-scaled fp16 x 2) on the top levels
    "bf16x3": {"chol_planes": 3},                    # image GEMMs, exact 8+8+8 split, six products
    "fp32": {"chol_3p_min": 0, "chol_fp32": 1},      # v_mfma_f32_32x32x2_f32 everywhere = the reference's precision
}


def fp64_chain(Hd):
    """gptq.py:318-320 in fp64 on the device (torch / hipSOLVER)."""
    H64 = Hd.double()
    L = torch.linalg.cholesky(H64)
    del H64
    Hi = torch.cholesky_inverse(L)
    del L
    return torch.linalg.cholesky(Hi, upper=True)


def u_errors(U, U64):
    """(max-norm error, worst row-wise error, worst relative error on the diagonal) of U against the fp64 chain."""
    d = (U.double() - U64).abs_()
    rowmax = U64.abs().max(dim=1).values
    out = (float(d.max() / U64.abs().max()), float((d.max(dim=1).values / rowmax).max()),
           float((d.diagonal() / U64.diagonal().abs()).max()))
    del d
    return out


def equilibrated_cond(Hd):
    """cond_2 of D^-1/2 Hd D^-1/2 (what the chain factorises, up to the power-of-two rounding of the scales)."""
    s = Hd.diagonal().double().rsqrt()
    E = Hd.double() * s[:, None] * s[None, :]
    ev = torch.linalg.eigvalsh(E)
    return float(ev[-1] / ev[0])
