#!/usr/bin/env python3
"""Practical MFMA peak calibration: hipBLASLt fp16 GEMM at SYRK-like shapes (X^T X with K = tokens)."""
import torch, time
dev = torch.device("cuda")
for (M, N, K) in [(8192, 8192, 65536), (8192, 8192, 8192), (14336, 14336, 16384)]:
    A = torch.randn(M, K, device=dev, dtype=torch.float16)
    B = torch.randn(N, K, device=dev, dtype=torch.float16)
    for _ in range(2):
        C = A @ B.t()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        C = A @ B.t()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"hipBLASLt NT {M}x{N}x{K}: {dt*1e3:.2f} ms  {2*M*N*K/dt/1e12:.0f} TFLOP/s")
