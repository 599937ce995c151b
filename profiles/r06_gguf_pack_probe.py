#!/usr/bin/env python3
"""The packer leg alone: Quantizer.quantize of a random-init Llama-3-8B (LAYERS blocks), then pack_gptq_into_gguf.convert under
GGUFWriter.LAZY_WORKERS = 1, 2, 3, 4 and as the tensor-by-tensor flow.  usage: [LAYERS=32] python profiles/r06_gguf_pack_probe.py"""
import os, sys, time, shutil, tempfile, hashlib
from pathlib import Path
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(R, "bench.py")); B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
from gptq_gguf_toolkit_amd.quantizer import Quantizer
from gptq_gguf_toolkit_amd.gguf_writer import GGUFWriter
from gptq_gguf_toolkit_amd.pack_gptq_into_gguf import convert
dev = torch.device("cuda:0")
wl = B.WORKLOADS["llama3-8b-model-q4k"]
cfg = dict(wl["model"]); cfg["num_hidden_layers"] = int(os.environ.get("LAYERS", 32))
model = B.build_model(cfg, dev)
root = "/dev/shm"
hf = tempfile.mkdtemp(prefix="gq_hf_", dir=root); sd = tempfile.mkdtemp(prefix="gq_sd_", dir=root)
try:
    model.save_pretrained(hf, safe_serialization=True)
    g = torch.Generator().manual_seed(1)
    nseq = int(os.environ.get("NSEQ", 8))
    data = [([], {"input_ids": torch.randint(0, cfg["vocab_size"], (1, 2048), generator=g)}) for _ in range(nseq)]
    q = B.QT[wl["q"]]
    qc = {k: q for k in ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "down_proj", "up_proj", "embed_tokens", "lm_head")}
    drv = Quantizer(model, data_loader=data, quantizable_modules=r".*layers.*((q|k|v|o|gate|up|down)_proj)$",
                    quantizer_kwargs=dict(B.QUANTIZER_KW, verbose=False), pre_block_modules=["model.embed_tokens"], block_modules="model.layers",
                    post_block_modules=["lm_head"], quant_non_block_modules=True, device=str(dev), save_dir=sd)
    t0 = time.perf_counter(); drv.quantize(qc); torch.cuda.synchronize(); print(f"quantize ({nseq} calibration sequences) {time.perf_counter()-t0:.2f} s", flush=True)
    del drv, model; torch.cuda.empty_cache()
    def sha(p):
        h = hashlib.sha256()
        with open(p, "rb") as f:
            for c in iter(lambda: f.read(1 << 26), b""): h.update(c)
        return h.hexdigest()[:12]
    for label, kw, workers in [("pipelined x1", dict(), 1), ("pipelined x2", dict(), 2), ("pipelined x3", dict(), 3), ("pipelined x4", dict(), 4), ("pipelined x6", dict(), 6),
                               ("tensor by tensor", dict(pipelined=False), 1), ("pipelined x3", dict(), 3)]:
        GGUFWriter.LAZY_WORKERS = workers
        GGUFWriter.LAZY_DEPTH = max(4, workers + 2)
        out = os.path.join(root, "gq_probe.gguf"); tm = {}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        convert(Path(hf), Path(sd), Path(out), "f16", vocab=False, timing=tm, **kw)
        dt = time.perf_counter() - t0
        print(f"{label:18s} {dt:6.2f} s  {os.path.getsize(out)/1e9:.2f} GB  sha {sha(out)}  " + " ".join(f"{k}={v:.2f}" for k, v in sorted(tm.items())), flush=True)
        os.remove(out)
finally:
    shutil.rmtree(hf, ignore_errors=True); shutil.rmtree(sd, ignore_errors=True)
