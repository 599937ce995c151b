"""Which hipBLASLt / Tensile kernel does torch pick for the SYRK-shaped fp16 GEMM (X^T X, T = 65536, C = 4096 / 14336)?
Run under `rocprofv3 --kernel-trace`: the Tensile kernel name spells the macro tile, the MFMA shape and the LDS settings."""
import time
import torch
for C in (4096, 14336):
    T = 65536
    X = torch.randn(T, C, device="cuda", dtype=torch.float16)
    for _ in range(2):
        H = X.T @ X
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        H = X.T @ X
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"C={C}: {dt * 1e3:.2f} ms, {2 * T * C * C / dt / 1e12:.0f} TFLOP/s (full product)")
