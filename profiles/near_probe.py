"""Column loop of one 4096 x 14336 Linear twice (for rocprofv3 --kernel-trace: per-kernel durations of the near / far
launches without event overhead).  usage: bash profiles/near_probe.sh"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptq_gguf_toolkit_amd import ops  # noqa: E402

R, C = 4096, 14336
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.randn(8192, C, device="cuda", generator=g).half()
H = torch.zeros(C, C, device="cuda")
ops.h_accumulate(H, X, 0.0, 2.0 / 8)
W = (torch.randn(R, C, device="cuda", generator=g) * 0.02)
U, _ = ops.h_prepare(H, W, 0.01)
ops.far_helper_enable(False)
for _ in range(2):
    Wf = W.clone()
    ops.gptq_quantize(Wf, U, 12, 128)
    torch.cuda.synchronize()
