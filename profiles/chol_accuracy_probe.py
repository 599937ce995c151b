"""Accuracy of U = chol_upper(H^-1) on an outlier-channel Hessian: split-bf16 GEMMs vs fp32-only GEMMs vs fp64."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptq_gguf_toolkit_amd import ops
torch.manual_seed(3)
C = 4096
sig = torch.exp(torch.randn(C, device="cuda") * 0.5)
sig[torch.randperm(C, device="cuda")[:8]] *= 20.0           # outlier channels
X = (torch.randn(4 * C, C, device="cuda") * sig).half()
H0 = torch.zeros(C, C, device="cuda")
ops.h_accumulate(H0, X, 0.0, 2.0 / 8)
W = torch.randn(64, C, device="cuda")
# the threshold is read once per process: run with GQ_CHOL_FP32=1 for the fp32-only factorisation
res = {}
U, flag = ops.h_prepare(H0.clone(), W.clone(), 0.01)
res["fp32 only" if os.environ.get("GQ_CHOL_FP32") else "split-bf16 (nodes >= %s)" % os.environ.get("GQ_CHOL_3B_MIN", "512")] = U.double()
Hd = H0.double(); Hd += 0.01 * Hd.diagonal().mean() * torch.eye(C, device="cuda", dtype=torch.float64)
ref = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(Hd)), upper=True)
for mode, U in res.items():
    print(mode, "max|U-ref|/max|ref| = %.3e" % ((U - ref).abs().max() / ref.abs().max()).item(),
          " resid |U^T U H - I|max = %.3e" % ((U.T @ U @ Hd) - torch.eye(C, device="cuda", dtype=torch.float64)).abs().max().item())
print("cond(H_damped) ~ %.2e" % (torch.linalg.eigvalsh(Hd)[-1] / torch.linalg.eigvalsh(Hd)[0]).item())
