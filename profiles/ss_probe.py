"""Scale-search kernel timing: lane-per-group (default) vs 8-lanes-per-group (GQ_SS_WIDE=1), isolated.
usage: python profiles/ss_probe.py   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gptq_gguf_toolkit_amd import ops

torch.manual_seed(0)
for rows in (1024, 4096, 14336):
    W = torch.randn(rows, 4096, device="cuda")
    for q in (12, 10):
        for env in ("0", "2", "1"):
            os.environ["GQ_SS_WIDE"] = env
            for _ in range(3): ops.scale_search(W[:, :256], q)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(64): ops.scale_search(W[:, (i % 16) * 256:(i % 16) * 256 + 256], q)
            e1.record(); torch.cuda.synchronize()
            print(f"rows={rows} q={q} { {'0': 'lane', '2': 'pair', '1': 'wide'}[env] }: {e0.elapsed_time(e1) / 64 * 1e3:.1f} us/launch")
os.environ.pop("GQ_SS_WIDE", None)
