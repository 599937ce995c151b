"""SYRK alone on the GPU with and without the K-split of the last round (GQ_SYRK_NOSPLIT=1): C=<width> python profiles/ksplit_probe.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptq_gguf_toolkit_amd import ops
C, T = int(os.environ.get("C", 14336)), 65536
X = (torch.randn(T, C, device="cuda") * torch.exp(torch.randn(C, device="cuda") * 0.5)).half()
H = torch.zeros(C, C, device="cuda")
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ops.h_accumulate(H, X, 0.5, 1e-5)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
nt = C // 128
print(f"C={C} T={T} nosplit={os.environ.get('GQ_SYRK_NOSPLIT')}: {dt*1e3:.2f} ms  {2.0*T*128*128*(nt*(nt+1)//2)/dt/1e12:.0f} TFLOP/s")
