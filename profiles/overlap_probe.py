#!/usr/bin/env python3
"""Can a 7168 factorisation hide behind the 14336 SYRK?  Both alone, then concurrently on two streams."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptq_gguf_toolkit_amd import ops
dev = torch.device("cuda")
C, Ch, nseq, L = 14336, 7168, 128, 2048
X = torch.randn(nseq * L, C, device=dev, dtype=torch.float16)
H = torch.zeros(C, C, device=dev)
Xs = (torch.randn(4 * Ch, Ch, device=dev) * torch.exp(torch.randn(Ch, device=dev) * 0.5)).half()
Hs0 = torch.zeros(Ch, Ch, device=dev)
ops.h_accumulate(Hs0, Xs, 0.0, 2.0 / 8)
W = torch.randn(256, Ch, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def syrk():
    ops.h_accumulate_grouped([H], [X], [0.0], [2.0 / nseq])
def chol():
    return ops.h_prepare(Hs0.clone(), W.clone(), 0.01)
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); syrk(); torch.cuda.synchronize(); ta = time.perf_counter() - t0
    torch.cuda.synchronize(); t0 = time.perf_counter(); chol(); torch.cuda.synchronize(); tb = time.perf_counter() - t0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e0, e1, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); s1.wait_event(e0); s2.wait_event(e0)
    with torch.cuda.stream(s1):
        syrk(); e1.record()
    with torch.cuda.stream(s2):
        chol(); e2.record()
    torch.cuda.synchronize(); tc = time.perf_counter() - t0
print(f"concurrent: SYRK done at {e0.elapsed_time(e1):.1f} ms, chol done at {e0.elapsed_time(e2):.1f} ms")
print(f"SYRK(14336) alone {ta*1e3:.1f} ms   chol(7168) alone {tb*1e3:.1f} ms   both concurrently {tc*1e3:.1f} ms")
