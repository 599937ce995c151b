#!/usr/bin/env python3
"""gq_h_prepare alone, resident sub-problem launches (chol_sub) against launch-by-launch, by workgroup count."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptq_gguf_toolkit_amd import ops
for C in [int(c) for c in os.environ.get("CS", "4096,14336").split(",")]:
    torch.manual_seed(0)
    X = (torch.randn(2 * C, C, device="cuda") * torch.exp(torch.randn(C, device="cuda") * 0.5)).half()
    H0 = torch.zeros(C, C, device="cuda"); ops.h_accumulate(H0, X, 0.0, 0.5); del X
    W0 = torch.randn(128, C, device="cuda")
    ref = None
    for label, kw in [("launches", dict(chol_sub=0))] + [(f"resident wgs={g}", dict(chol_sub=16, chol_sub_wgs=g)) for g in (16, 32, 48, 64, 96, 128)]:
        with ops.options(**kw):
            ts = []
            for it in range(4):
                H = H0.clone(); W = W0.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
                U, flag = ops.h_prepare(H, W, 0.01)
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        if ref is None: ref = U.clone()
        print(f"C={C} {label:22s} {min(ts)*1e3:7.2f} ms  identical={torch.equal(ref, U)} flag={int(flag.item())}", flush=True)
