"""Host time of one HF decoder-layer forward with and without the quantizer's hooks (is forward #1 host-bound?).
usage (GPU box): python profiles/hook_overhead_probe.py"""
import cProfile
import os
import pstats
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gptq_gguf_toolkit_amd.block_schedule import BlockSchedule
from gptq_gguf_toolkit_amd.gptq import GPTQ
from gptq_gguf_toolkit_amd.model_utils import LINEAR_LAYERS, select_layers

dev = torch.device("cuda:0")
cfg = dict(bench.WORKLOADS["llama3-8b-model-q4k"]["model"], num_hidden_layers=1)
model = bench.build_model(cfg, dev)
block = model.model.layers[0]
L, n = 2048, 64
h = torch.randn(1, L, 4096, device=dev, dtype=torch.bfloat16)
pos = torch.arange(L, device=dev).unsqueeze(0)
pe = model.model.rotary_emb(h, pos)
kw = dict(position_ids=pos, position_embeddings=pe, attention_mask=None, use_cache=False)


def run(tag, sched=None):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        for _ in range(n):
            block(h, **kw)
            if sched is not None:
                sched.sample_done()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{tag:28s} host {1e3 * (t1 - t0) / n:6.3f} ms/sample, host+drain {1e3 * (t2 - t0) / n:6.3f} ms/sample")


run("warm")
run("no hooks")
layers = select_layers(model, "model.layers.0.", r".*layers.*((q|k|v|o|gate|up|down)_proj)$", LINEAR_LAYERS)
sched = BlockSchedule(layers, lambda l, nm: GPTQ(l, rel_damp=0.01, block_size=128))
hooks = [l.register_forward_hook(sched.hook(nm)) for nm, l in layers.items()]
run("hooks (zero copy)", sched)
run("hooks (zero copy) again", sched)
pr = cProfile.Profile()
pr.enable()
run("hooks under cProfile", sched)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
