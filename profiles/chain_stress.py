"""Stress of the asm-scheduled far-update kernel: look-ahead (chained GEMM) vs per-block updates must be
bit-identical for several shapes and types while another stream keeps the GPU busy (varies the timing of LDS
traffic and barriers).  usage: python profiles/chain_stress.py   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gptq_gguf_toolkit_amd import ops
torch.manual_seed(0)
side = torch.cuda.Stream()
A = torch.randn(8192, 8192, device="cuda", dtype=torch.float16)
bad = 0
for it, (R, C, q) in enumerate([(128, 2048, 12), (384, 5120, 10), (1024, 4096, 14), (256, 14336, 12), (640, 3072, 13),
                                (4096, 4096, 12), (128, 2048, 11), (2048, 6144, 12)] * 2):
    W0 = (torch.randn(R, C, device="cuda") * 0.02).half().float()
    U = torch.triu(torch.randn(C, C, device="cuda") * (0.3 / C ** 0.5), 1) + torch.diag(0.5 + torch.rand(C, device="cuda"))
    outs = []
    for env in (None, "1"):
        if env: os.environ["GQ_NO_LOOKAHEAD"] = env
        else: os.environ.pop("GQ_NO_LOOKAHEAD", None)
        with torch.cuda.stream(side):
            for _ in range(6 + it % 5): B = A @ A  # background load
        W = W0.clone()
        outs.append((W,) + tuple(ops.gptq_quantize(W, U, q, 128)))
        torch.cuda.synchronize()
    os.environ.pop("GQ_NO_LOOKAHEAD", None)
    same = all(torch.equal(a.view(torch.uint8), b.view(torch.uint8)) for a, b in zip(*outs))
    bad += not same
    print(f"R={R} C={C} q={q}: {'identical' if same else 'DIFFERENT'}")
print("FAILURES:", bad)
sys.exit(1 if bad else 0)
