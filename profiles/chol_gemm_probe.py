#!/usr/bin/env python3
"""gq_chol_gemm (the image GEMMs of the Cholesky chain) alone: accuracy against fp64 on small shapes in every mode,
then time and TFLOP/s (fp32-equivalent: 2 M N K_effective) at the shapes of a 14336-wide factorisation."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptq_gguf_toolkit_amd import ops

dev = "cuda"
torch.manual_seed(0)


def ref(C0, A, B, trans_b, mode, kr, lower):
    A64, B64 = A.double(), B.double()
    if kr == 3:
        A64 = torch.tril(A64)
    if kr == 1:
        B64 = torch.tril(B64)  # B [N,K] lower
    if kr == 2:
        B64 = torch.tril(B64)  # B [K,N] lower: k >= n
    P = A64 @ (B64.T if trans_b else B64)
    if mode == 0:
        out = C0.double() - P
    elif mode == 1:
        out = P
    else:
        out = -P
    return out


def poison(T, which):
    """128x128 blocks beyond the block diagonal of a triangular operand hold garbage: they must never be read."""
    n = T.shape[0] // 128
    for i in range(n):
        for j in range(n):
            if (which == "upper" and j > i):
                T[128 * i:128 * i + 128, 128 * j:128 * j + 128] = float("nan")
    return T


def check(M, N, K, trans_b, mode, kr, lower, planes):
    A = torch.randn(M, K, device=dev) * torch.exp(torch.randn(M, 1, device=dev) * 2)
    B = (torch.randn(N, K, device=dev) if trans_b else torch.randn(K, N, device=dev)) * 0.1
    if kr == 3:
        A = poison(torch.tril(A), "upper")
    if kr in (1, 2):
        B = poison(torch.tril(B), "upper")
    if lower:
        B = A
    C0 = torch.randn(M, N, device=dev)
    Cm = C0.clone()
    ops.chol_gemm(Cm, A, B, trans_b, mode, kr, lower, planes)
    torch.cuda.synchronize()
    R = ref(C0, torch.nan_to_num(A), torch.nan_to_num(B), trans_b, mode, kr, lower)
    if lower:
        msk = torch.ones(M // 256, N // 256, device=dev).tril().repeat_interleave(256, 0).repeat_interleave(256, 1).bool()
    else:
        msk = torch.ones(M, N, device=dev).bool()
    absA, absB = torch.nan_to_num(A).abs().double(), torch.nan_to_num(B).abs().double()
    bound = absA @ (absB.T if trans_b else absB) + C0.abs().double()
    err = ((Cm.double() - R).abs() / bound)[msk].max().item()
    # fp32 sgemm reference error on the same problem
    P32 = (torch.nan_to_num(A) @ (torch.nan_to_num(B).T if trans_b else torch.nan_to_num(B)))
    tri = ref(torch.zeros_like(C0), torch.nan_to_num(A), torch.nan_to_num(B), trans_b, 1, kr, lower)
    print(f"M={M} N={N} K={K} tb={int(trans_b)} mode={mode} kr={kr} lower={int(lower)} planes={planes}: "
          f"max componentwise err / (|A||B|) = {err:.2e}  nan={int(torch.isnan(Cm[msk]).any())}")
    return err


if os.environ.get("CHECK", "1") == "1":
    worst = 0.0
    for planes in (3, 2):
        for (tb, mode, kr, lower) in [(True, 1, 1, False), (True, 0, 0, True), (False, 1, 2, False), (False, 2, 3, False),
                                      (True, 1, 0, False), (False, 0, 0, False)]:
            for (M, N, K) in [(512, 512, 512), (1024, 768, 768) if kr == 0 else (1024, 1024, 1024)]:
                if lower and M != N:
                    N = M
                if kr in (1, 2) and N != K:
                    K = N
                if kr == 3 and M != K:
                    K = M
                worst = max(worst, check(M, N, K, tb, mode, kr, lower, planes))
    print("worst", worst)

if os.environ.get("TIME", "1") == "1":
    for planes in (3, 2):
        for n in (7168, 3584, 1792):
            A = torch.randn(n, n, device=dev)
            X = torch.tril(torch.randn(n, n, device=dev))
            if os.environ.get("DATA") == "zeros":  # no power cap: the kernel's structural rate
                A.zero_(); X.zero_()
            Cm = torch.zeros(n, n, device=dev)
            for name, args, flops in [("G1 A X^T kr1", (Cm, A, X, True, 1, 1, False), n ** 3),
                                      ("G2 syrk lower", (Cm, A, A, True, 0, 0, True), n ** 3),
                                      ("G3 A X kr2", (Cm, A, X, False, 1, 2, False), n ** 3),
                                      ("G4 -X A kr3", (Cm, X, A, False, 2, 3, False), n ** 3)]:
                ops.chol_gemm(*args, planes)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    ops.chol_gemm(*args, planes)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 3
                print(f"planes={planes} n={n} {name}: {dt*1e3:.3f} ms incl. splits = {flops/dt/1e12:.1f} TFLOP/s fp32-equivalent")
