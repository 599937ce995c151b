"""HBM-bound codec kernels (K7 dequantize, K8 RTN, K9-13 pack) alone on the GPU: time and achieved GB/s against the
algorithmic bytes of SURVEY 8d.  usage: python profiles/codec_probe.py   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gptq_gguf_toolkit_amd import ops
TS = {10: 84, 11: 110, 12: 144, 13: 176, 14: 210}
NAME = {10: "Q2_K", 11: "Q3_K", 12: "Q4_K", 13: "Q5_K", 14: "Q6_K"}
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
torch.manual_seed(0)
R, C = 4096, 14336
W = (torch.randn(R, C, device="cuda") * 0.02).half()
for t in (12, 10, 14, 11, 13):
    q, d, s, dmin, m = ops.rtn_quantize(W, t)
    G = 32 if t in (12, 13) else 16
    aux = R * (C // 256) * 4 + 2 * R * (C // G)                     # d, dmin fp16 + s, m bytes
    tr = timeit(lambda: ops.rtn_quantize(W, t), 5)
    td = timeit(lambda: ops.dequantize(t, q, d, s, dmin, m, torch.float16))
    tp = timeit(lambda: ops.pack(t, q, d, s, dmin, m))
    b_rtn = R * C * (2 + 1) + aux                                   # read fp16 W, write ints + scales
    b_deq = R * C * (1 + 2) + aux                                   # read ints + scales, write fp16
    b_pack = R * C * 1 + aux + R * (C // 256) * TS[t]               # read ints + scales, write blocks
    print(f"{NAME[t]} {R}x{C}: rtn_quantize {tr*1e3:7.3f} ms {b_rtn/tr/1e9:7.0f} GB/s | dequantize {td*1e6:6.1f} us {b_deq/td/1e9:6.0f} GB/s"
          f" | pack {tp*1e6:6.1f} us {b_pack/tp/1e9:6.0f} GB/s")
