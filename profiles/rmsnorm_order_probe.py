"""Is gq_fwd_rmsnorm_ordered bit-identical to HF's LlamaRMSNorm on this PyTorch?  (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformers.models.llama.modeling_llama import LlamaRMSNorm  # noqa: E402
from gptq_gguf_toolkit_amd import ops  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
for dt in (torch.bfloat16, torch.float16):
    for shape in ((1, 2048, 4096), (4, 2048, 4096), (1, 2048, 2048), (1, 2048, 5120), (1, 4096, 8192), (3, 77, 512), (1, 1, 4096), (2, 300, 14336)):
        C = shape[-1]
        x = (torch.randn(shape, device="cuda", generator=g) * torch.exp(torch.randn(C, device="cuda", generator=g))).to(dt)
        m = LlamaRMSNorm(C, 1e-5).cuda().to(dt)
        with torch.no_grad():
            m.weight.copy_((1 + 0.2 * torch.randn(C, device="cuda", generator=g)).to(dt))
            want = m(x)
            var = x.float().pow(2).mean(-1, keepdim=True)
        got = ops.fwd_rmsnorm_ordered(x, m.weight.data, 1e-5)
        free = ops.fwd_rmsnorm(x, m.weight.data, 1e-5)
        print(dt, shape, "ordered equal:", torch.equal(got, want), " mismatches:", int((got != want).sum()),
              " (free-order kernel mismatches:", int((free != want).sum()), ")")
