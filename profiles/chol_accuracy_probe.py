"""Accuracy of U = chol_upper(H^-1) on an outlier-channel Hessian: split-bf16 GEMMs vs fp32-only GEMMs vs fp64."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptq_gguf_toolkit_amd import ops
torch.manual_seed(3)
C = int(os.environ.get("C", "4096"))
sig = torch.exp(torch.randn(C, device="cuda") * float(os.environ.get("SPREAD", "0.5")))
sig[torch.randperm(C, device="cuda")[:8]] *= float(os.environ.get("OUTLIER", "20"))    # outlier channels
if float(os.environ.get("SPREAD", "0.5")) > 1.0:
    # channel scales beyond the fp16 range: build H = S (Z^T Z) S in fp32 directly
    Z = torch.randn(4 * C, C, device="cuda").half()
    H0 = torch.zeros(C, C, device="cuda")
    ops.h_accumulate(H0, Z, 0.0, 2.0 / 8)
    H0 = H0 * sig[:, None] * sig[None, :]
    H0 = (H0 + H0.T) * 0.5
    del Z
else:
    X = (torch.randn(4 * C, C, device="cuda") * sig).half()
    H0 = torch.zeros(C, C, device="cuda")
    ops.h_accumulate(H0, X, 0.0, 2.0 / 8)
W = torch.randn(64, C, device="cuda")
# the threshold is read once per process: run with GQ_CHOL_FP32=1 for the fp32-only factorisation
res = {}
U, flag = ops.h_prepare(H0.clone(), W.clone(), 0.01)
tag = "fp32 only" if os.environ.get("GQ_CHOL_FP32") else "split-bf16 (nodes >= %s)" % os.environ.get("GQ_CHOL_3B_MIN", "1024")
tag += ", images: " + ("off" if os.environ.get("GQ_CHOL_3P_MIN") == "0" else ("bf16x3" if os.environ.get("GQ_CHOL_BF16X3") else "fp16x2") + " >= " + os.environ.get("GQ_CHOL_3P_MIN", "1792")) + (", no equilibration" if os.environ.get("GQ_CHOL_NO_EQUIL") else "")
res[tag] = U.double()
Hd = H0.double(); Hd += 0.01 * Hd.diagonal().mean() * torch.eye(C, device="cuda", dtype=torch.float64)
ref = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(Hd)), upper=True)
for mode, U in res.items():
    d = (U - ref).abs()
    print(mode, "C=%d flag=%d" % (C, int(flag.item())),
          "rowwise max|dU_i|/max|ref_i| = %.3e" % (d.max(dim=1).values / ref.abs().max(dim=1).values).max().item(),
          "diag rel = %.3e" % (d.diagonal() / ref.diagonal().abs()).max().item(),
          "max|U-ref|/max|ref| = %.3e" % ((U - ref).abs().max() / ref.abs().max()).item(),
          " resid |U^T U H - I|max = %.3e" % ((U.T @ U @ Hd) - torch.eye(C, device="cuda", dtype=torch.float64)).abs().max().item())
print("cond(H_damped) ~ %.2e" % (torch.linalg.eigvalsh(Hd)[-1] / torch.linalg.eigvalsh(Hd)[0]).item())
