"""torch (rocBLAS / hipBLASLt sgemm, TF32 off as the reference sets it) on the shapes of the far trailing update
W[:, S1:] -= Err[R, 1024] @ U[S0:S1, S1:], R = 4096: the first (N = 13312) and a middle (N = 7168) super-block."""
import time
import torch
torch.backends.cuda.matmul.allow_tf32 = False
R, K = 4096, 1024
for N in (13312, 7168, 1024):
    W = torch.randn(R, N, device="cuda")
    E = torch.randn(R, K, device="cuda")
    U = torch.randn(K, N, device="cuda")
    for _ in range(3):
        W.addmm_(E, U, alpha=-1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        W.addmm_(E, U, alpha=-1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"N={N}: {dt * 1e3:.3f} ms, {2 * R * N * K / dt / 1e12:.1f} TFLOP/s")
