#!/usr/bin/env python3
"""A/B of the column loop alone on the GPU (down_proj 4096 x 14336 and the stacked gate/up 28672 x 4096, Q4_K), options via
GQ_OPTIONS: ms per loop (best of 5) and a hash of the results."""
import hashlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptq_gguf_toolkit_amd import ops
torch.manual_seed(0)
dev = "cuda"
for R, C in ((4096, 14336), (28672, 4096), (4096, 4096)):
    W = (torch.randn(R, C, device=dev) * 0.02).half().float()
    X = torch.randn(8192, C, device=dev, dtype=torch.float16)
    H = torch.zeros(C, C, device=dev)
    ops.h_accumulate(H, X, 0.0, 2.0 / 4)
    U, _ = ops.h_prepare(H, W.clone(), 0.01)
    best = 1e9
    for it in range(5):
        Wf = W.clone()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = ops.gptq_quantize(Wf, U, 12, 128)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    h = hashlib.sha256()
    for t in (Wf,) + tuple(res):
        h.update(t.cpu().numpy().tobytes())
    print(f"{R}x{C}: {best * 1e3:.3f} ms  hash {h.hexdigest()[:12]}  [{os.environ.get('GQ_OPTIONS', '')}]")
