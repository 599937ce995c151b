#!/usr/bin/env python3
"""gq_h_prepare alone (no other streams): per-tag breakdown for a few sizes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptq_gguf_toolkit_amd import ops, _cabi
for C in [int(c) for c in os.environ.get("CS", "4096,14336").split(",")]:
    T = 4 * C
    X = (torch.randn(T, C, device="cuda") * torch.exp(torch.randn(C, device="cuda") * 0.5)).half()
    H0 = torch.zeros(C, C, device="cuda")
    ops.h_accumulate(H0, X, 0.0, 2.0 / 8)
    del X
    W = torch.randn(256, C, device="cuda")
    for it in range(2):
        H = H0.clone(); torch.cuda.synchronize()
        _cabi.prof_enable(None); t0 = time.perf_counter()
        U, flag = ops.h_prepare(H, W.clone(), 0.01)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        bd = _cabi.prof_collect(); _cabi.prof_enable([])
    print(f"C={C}: {dt*1e3:.1f} ms flag={int(flag.item())}  " + "  ".join(f"{k}={v[0]:.1f}ms/{v[1]}" for k, v in sorted(bd.items(), key=lambda kv: -kv[1][0])))
