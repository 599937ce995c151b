#!/usr/bin/env python3
"""Runs ONLY the grouped Hessian SYRK on the bench shapes (for rocprofv3 --pmc passes)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptq_gguf_toolkit_amd import ops

nseq, L = int(os.environ.get("NSEQ", 32)), 2048
dev = torch.device("cuda")
Cs = [int(c) for c in os.environ.get("CS", "4096,4096,4096,14336").split(",")]
zeros = os.environ.get("DATA", "random") == "zeros"  # all-zero operands: no power cap (the structural ceiling of the kernel)
X = [torch.zeros(nseq * L, C, device=dev, dtype=torch.float16) if zeros else
     torch.randn(nseq * L, C, device=dev, dtype=torch.float16) for C in Cs]
H = [torch.zeros(C, C, device=dev) for C in Cs]
for it in range(int(os.environ.get("ITERS", 2))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ops.h_accumulate_grouped(H, X, [0.0] * len(Cs), [2.0 / nseq] * len(Cs))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    fl = sum(2.0 * nseq * L * 128 * 128 * ((C // 128) * (C // 128 + 1) // 2) for C in Cs)
    print(f"iter {it}: {dt*1e3:.2f} ms  {fl/dt/1e12:.1f} TFLOP/s (incl. transpose)")
