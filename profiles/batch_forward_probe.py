"""GPU time per calibration sequence of one Llama-3-8B decoder layer forward, batch 1 / 2 / 4 / 8 sequences per call.
usage (GPU box): python profiles/batch_forward_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda:0")
cfg = dict(bench.WORKLOADS["llama3-8b-model-q4k"]["model"], num_hidden_layers=1)
model = bench.build_model(cfg, dev)
block = model.model.layers[0]
L = 2048
for b in (1, 2, 4, 8, 16):
    h = torch.randn(b, L, 4096, device=dev, dtype=torch.bfloat16)
    pos = torch.arange(L, device=dev).unsqueeze(0)
    pe = model.model.rotary_emb(h, pos)
    kw = dict(position_ids=pos, position_embeddings=pe, attention_mask=None, use_cache=False)
    n = 64 // b
    with torch.no_grad():
        for _ in range(2):
            block(h, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            block(h, **kw)
        torch.cuda.synchronize()
    print(f"batch {b:2d}: {1e3 * (time.perf_counter() - t0) / (n * b):6.3f} ms per sequence")
