#!/usr/bin/env python3
"""The critical chain of a Llama-3-8B block alone: gq_h_prepare(14336) + gq_gptq_quantize(4096 x 14336, Q4_K)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptq_gguf_toolkit_amd import ops, _cabi
Q4_K = 12
R, C = int(os.environ.get("R", 4096)), int(os.environ.get("C", 14336))
T = 2 * C
X = (torch.randn(T, C, device="cuda") * torch.exp(torch.randn(C, device="cuda") * 0.5)).half()
H0 = torch.zeros(C, C, device="cuda")
ops.h_accumulate(H0, X, 0.0, 2.0 / 8)
del X
W0 = torch.randn(R, C, device="cuda") * 0.02
for it in range(2):
    H = H0.clone(); W = W0.clone(); torch.cuda.synchronize()
    _cabi.prof_enable([] if os.environ.get("PROF") == "0" else None); t0 = time.perf_counter()
    U, flag = ops.h_prepare(H, W, 0.01)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    out = ops.gptq_quantize(W, U, Q4_K, 128)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    bd = _cabi.prof_collect(); _cabi.prof_enable([])
print(f"R={R} C={C}: prepare {1e3*(t1-t0):.1f} ms  quantize {1e3*(t2-t1):.1f} ms  flag={int(flag.item())}")
for k, v in sorted(bd.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:22s} {v[0]:8.2f} ms / {v[1]:4d} launches = {1e3*v[0]/v[1]:8.1f} us each")
