"""The reference's own chain on this GPU through torch (hipSOLVER / MAGMA under it), next to gq_h_prepare:
U = cholesky(cholesky_inverse(cholesky(H)), upper=True) (gptq.py:319-320), C = 4096 and 14336, fp32."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gptq_gguf_toolkit_amd import ops
for C in (4096, 14336):
    X = (torch.randn(2 * C, C, device="cuda") * torch.exp(torch.randn(C, device="cuda") * 0.5)).half()
    H = torch.zeros(C, C, device="cuda")
    ops.h_accumulate(H, X, 0.0, 2.0 / 8)
    del X
    H.diagonal().add_(0.01 * H.diagonal().mean())
    W = torch.randn(256, C, device="cuda")
    res = {}
    for name in ("torch", "gq"):
        best = 1e9
        for it in range(3):
            Hc = H.clone()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if name == "torch":
                U = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(Hc)), upper=True)
            else:
                U, _ = ops.h_prepare(Hc, W.clone(), 0.0)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        res[name] = (best, U)
    d = (res["torch"][1] - res["gq"][1]).abs().max() / res["torch"][1].abs().max()
    print(f"C={C}: torch chain {res['torch'][0] * 1e3:.1f} ms, gq_h_prepare {res['gq'][0] * 1e3:.1f} ms, max|dU|/max|U| = {d:.2e}")
