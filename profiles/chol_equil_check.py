"""Is U bit-identical with and without the power-of-two equilibration when only scale-invariant kernels run
(GQ_CHOL_3P_MIN=0: fp32 + exact-split bf16 GEMMs)?  Run twice (GQ_CHOL_NO_EQUIL unset / set) and compare the dumps."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptq_gguf_toolkit_amd import ops
torch.manual_seed(5)
C = int(os.environ.get("C", "4096"))
sig = torch.exp(torch.randn(C, device="cuda") * 0.5)
sig[torch.randperm(C, device="cuda")[:8]] *= 20.0
X = (torch.randn(2 * C, C, device="cuda") * sig).half()
H = torch.zeros(C, C, device="cuda")
ops.h_accumulate(H, X, 0.0, 2.0 / 8)
U, flag = ops.h_prepare(H, torch.randn(64, C, device="cuda"), 0.01)
out = os.environ["OUT"]
if os.path.exists(out):
    V = torch.load(out).cuda()
    print("C=%d identical=%s  max|dU|/max|U| = %.3e" % (C, bool(torch.equal(U, V)), ((U - V).abs().max() / V.abs().max()).item()))
else:
    torch.save(U.cpu(), out)
    print("saved", out)
