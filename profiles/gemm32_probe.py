#!/usr/bin/env python3
"""fp32 MFMA GEMM (gq_trailing_update) throughput at a few shapes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptq_gguf_toolkit_amd import ops
for M, N, K in ((4096, 4096, 128), (4096, 2048, 128), (14336, 4096, 128), (8192, 8192, 1024), (7168, 7168, 7168), (4096, 14336, 128)):
    A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda"); C = torch.randn(M, N, device="cuda")
    ops.trailing_update(C, A, B); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 5
    for _ in range(n): ops.trailing_update(C, A, B)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"M={M} N={N} K={K}: {dt*1e6:.0f} us  {2.0*M*N*K/dt/1e12:.1f} TFLOP/s")
