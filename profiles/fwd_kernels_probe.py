"""Times the three forward kernels against the HF eager modules at the Llama-3-8B block shape (one 2048-token sequence
and four), HIP events around 50 calls each.  Usage (GPU box): python profiles/fwd_kernels_probe.py"""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformers.models.llama.modeling_llama import LlamaRMSNorm, apply_rotary_pos_emb  # noqa: E402
from gptq_gguf_toolkit_amd import ops  # noqa: E402


def timed(fn, n=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3  # us


out = {}
for B in (1, 4):
    L, C, H, Hkv, D, I = 2048, 4096, 32, 8, 128, 14336
    dt = torch.bfloat16
    x = torch.randn(B, L, C, device="cuda").to(dt)
    norm = LlamaRMSNorm(C, 1e-5).cuda().to(dt)
    q = torch.randn(B, L, H, D, device="cuda").to(dt)
    k = torch.randn(B, L, Hkv, D, device="cuda").to(dt)
    cos = torch.randn(B, L, D, device="cuda").to(dt)
    sin = torch.randn(B, L, D, device="cuda").to(dt)
    g = torch.randn(B, L, I, device="cuda").to(dt)
    u = torch.randn(B, L, I, device="cuda").to(dt)
    with torch.no_grad():
        r = {
            "rmsnorm_eager_us": timed(lambda: norm(x)), "rmsnorm_hip_us": timed(lambda: ops.fwd_rmsnorm(x, norm.weight.data, 1e-5)),
            "rope_eager_us": timed(lambda: apply_rotary_pos_emb(q.transpose(1, 2), k.transpose(1, 2), cos, sin)),
            "rope_hip_us": timed(lambda: (ops.fwd_rope(q, cos, sin), ops.fwd_rope(k, cos, sin))),
            "silu_mul_eager_us": timed(lambda: torch.nn.functional.silu(g) * u), "silu_mul_hip_us": timed(lambda: ops.fwd_silu_mul(g, u)),
        }
    r["rmsnorm_hip_GBps"] = 2 * 2 * x.numel() / r["rmsnorm_hip_us"] / 1e3
    r["silu_mul_hip_GBps"] = 3 * 2 * g.numel() / r["silu_mul_hip_us"] / 1e3
    r["rope_hip_GBps"] = 2 * 2 * (q.numel() + k.numel()) / r["rope_hip_us"] / 1e3
    out[f"batch{B}"] = {k_: round(v, 1) for k_, v in r.items()}
print(json.dumps(out, indent=1))
