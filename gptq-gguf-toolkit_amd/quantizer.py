"""Model-walk driver: drop-in for the reference's quant/gptq/src/quantizer.py `Quantizer`.

Same constructor, same walk (embed -> blocks in order -> lm_head), same hook protocol
(forward hooks feed inp[0] of every regex-matched Linear to its handle), same data.pth
schema (quantizer.py:268-275).  MI355X-first differences, none of which changes results:
  * Linears of a block that are fed by the SAME input tensor (q/k/v, gate/up) share one
    Hessian accumulation and one all-reduce instead of 3/2 identical ones;
  * with world_size > 1 the Linears of a block are assigned to owner ranks (LPT on
    R*C*(C+128)) and quantize concurrently; the 5 result tensors are broadcast from the
    owner (the reference computes everything on rank 0, gptq.py:158);
  * data.pth is written by rank 0 only (the reference lets every rank write the same file).
"""
import os
from typing import Any, Dict, Iterable, List, Optional

import torch
import torch.nn as nn

from . import dist_utils
from . import ops as _ops
from .gptq import GPTQ
from .model_utils import ForwardInterrupt, InputCollector, LINEAR_LAYERS, _to, select_layers
from .quant_utils import GGML_QUANT_SIZES, GGMLQuantizationType, dequantize_linear_weight


def _first(x):
    return x[0] if isinstance(x, (tuple, list)) else x


class Quantizer:
    def __init__(self, model: nn.Module, data_loader: Iterable, quantizable_modules: str,
                 quantizer_kwargs: Dict[str, Any], pre_block_modules: List[str], post_block_modules: List[str],
                 block_modules: str, save_dir: str, quant_non_block_modules: bool = False,
                 device: Optional[torch.device] = None, cpu_offload_modules: bool = False,
                 cpu_offload_activations: bool = False, verbose: bool = False, non_block_fp32: bool = False) -> None:
        self.model = model
        self.data_loader = data_loader
        self.quantizable_modules = quantizable_modules
        self.quantizer_kwargs = quantizer_kwargs
        self.pre_block_modules = pre_block_modules
        self.post_block_modules = post_block_modules
        self.block_modules = block_modules
        self.device = device
        self.cpu_offload_modules = cpu_offload_modules
        self.cpu_offload_activations = cpu_offload_activations
        self.quant_non_block_modules = quant_non_block_modules
        self.verbose = verbose
        self.save_dir = save_dir
        self.non_block_fp32 = non_block_fp32

    # ------------------------------------------------------------------ walk
    @torch.no_grad()
    def quantize(self, quant_config: Dict[str, GGMLQuantizationType]) -> None:
        device = self.device or next(self.model.parameters()).device
        blocks = self.model.get_submodule(self.block_modules)
        pre_blocks = [(n, self.model.get_submodule(n)) for n in self.pre_block_modules]
        post_blocks = [(n, self.model.get_submodule(n)) for n in self.post_block_modules]
        blocks[0] = blocks[0].to(device)
        for _, m in pre_blocks:
            m.to(device)
        use_cache = getattr(self.model.config, "use_cache", None)
        if use_cache is not None:
            self.model.config.use_cache = False
        # capture the inputs of block 0 for every calibration sample (quantizer.py:78-89)
        blocks[0] = InputCollector(blocks[0], cpu_offload=self.cpu_offload_activations)
        for inp_args, inp_kwargs in self.data_loader:
            try:
                self.model(*_to(inp_args, device=device), **_to(inp_kwargs, device=device))
            except ForwardInterrupt:
                pass
        input_args, input_kwargs = blocks[0].input_args, blocks[0].input_kwargs
        blocks[0] = blocks[0].module
        dist_utils.barrier()

        if self.quant_non_block_modules:
            for name, module in pre_blocks:
                self._quant_and_save_non_block(name, module.to(device), quant_config)
        if self.cpu_offload_modules:
            for _, m in pre_blocks:
                m.cpu()

        for block_id, block in enumerate(blocks):
            if self.verbose:
                dist_utils.print_on_main(f"Processing {self.block_modules} {block_id}/{len(blocks)}.")
            block = block.to(device)
            prefix = f"{self.block_modules}.{block_id}."
            layers = select_layers(self.model, prefix, self.quantizable_modules, LINEAR_LAYERS)
            handles, hooks, seen = self._prepare_hooks_and_handles(layers)
            for a, kw in zip(input_args, input_kwargs):  # forward #1: Hessians (quantizer.py:150-151)
                seen.clear()
                block(*_to(a, device=device), **_to(kw, device=device))
            seen.clear()
            for h in hooks.values():
                h.remove()
            dist_utils.barrier()
            self._quant_group(handles, quant_config)
            for a, kw in zip(input_args, input_kwargs):  # forward #2: propagate (quantizer.py:161-172)
                out = _first(block(*_to(a, device=device), **_to(kw, device=device)))
                if self.cpu_offload_activations:
                    out = out.cpu()
                if len(a) > 0:
                    a[0].data = out
                elif "hidden_states" in kw:
                    kw["hidden_states"] = out
                else:
                    raise ValueError("Unsupported block input format.")
            if self.cpu_offload_modules:
                block.cpu()
            del handles, hooks

        if self.quant_non_block_modules:
            for name, module in post_blocks:
                self._quant_and_save_non_block(name, module.to(device), quant_config)
        if use_cache is not None:
            self.model.config.use_cache = use_cache
        dist_utils.barrier()

    # --------------------------------------------------------- hooks / handles
    def _create_handle(self, layer, name: str = "") -> GPTQ:
        # MoE expert Linears ("...experts.<e>.w1"): the router may send them no calibration token at all, and a
        # data-dependent number per rank -- the handle then falls back to H = I and reduces H sample-weighted
        # (the reference asserts, gptq.py:126, and averages unweighted)
        return GPTQ(layer, allow_no_samples=".experts." in f".{name}.", **self.quantizer_kwargs)

    def _prepare_hooks_and_handles(self, layers: Dict[str, nn.Module]):
        handles: Dict[str, GPTQ] = {n: self._create_handle(l, n) for n, l in layers.items()}
        hooks = {}
        seen: Dict[Any, Any] = {}  # per block call: input identity -> (leader handle, tensor kept alive)

        def make_hook(name):
            def _hook(_, inp, out):
                x = inp[0]
                h = handles[name]
                key = (x.data_ptr(), tuple(x.shape), tuple(x.stride()), x.dtype, x._version)
                if key in seen and seen[key][0].d_col == h.d_col and seen[key][0] is not h:
                    leader = seen[key][0]
                    assert h.shared_H_with in (None, leader), "input sharing pattern changed between samples"
                    assert h.H is None, "handle switched from own Hessian to a shared one"
                    h.shared_H_with = leader
                    leader._has_followers = True
                else:
                    assert h.shared_H_with is None, "input sharing pattern changed between samples"
                    seen[key] = (h, x)
                h.update(x)
            return _hook

        for name, layer in layers.items():
            hooks[name] = layer.register_forward_hook(make_hook(name))
        return handles, hooks, seen

    # ------------------------------------------------------------- quantize
    def _save(self, name, q_type, qweight, d, s, dmin, m):
        if not dist_utils.is_main():
            return
        os.makedirs(os.path.join(self.save_dir, name), exist_ok=True)
        torch.save({"q_type": int(q_type), "qweight": qweight.cpu(), "super_group_scale": d.cpu(),
                    "super_group_zero": dmin.cpu(), "group_scale_quant": s.cpu(), "group_zero_quant": m.cpu()},
                   os.path.join(self.save_dir, name, "data.pth"))

    def _quant_group(self, handles: Dict[str, GPTQ], quant_config: Dict[str, GGMLQuantizationType]):
        world, rank = dist_utils.get_world_size(), dist_utils.get_rank()
        qtypes = {n: quant_config.get(n.split(".")[-1], GGMLQuantizationType.Q4_K) for n in handles}
        if world > 1:
            costs = {n: float(h.d_row) * h.d_col * (h.d_col + 128) for n, h in handles.items()}
            # a matrix that outweighs a fair share is quantized by every rank on its own rows (U replicated);
            # the rest are handed out whole
            split = {n for n in dist_utils.row_split_names(costs, world) if not handles[n].act_order}
            for n in split:
                handles[n].row_split = True
            for n, r in dist_utils.assign_owners({n: c for n, c in costs.items() if n not in split}, world).items():
                handles[n].owner_rank = r
        # phase 0: one all-reduce per distinct Hessian, same order on every rank
        for h in handles.values():
            h.sync_hessian()
        # phase 1: owners run prepare + column loop, no communication in between
        results = {}
        order = sorted(handles, key=lambda n: not getattr(handles[n], "row_split", False))  # split matrices first
        for n in order:
            h = handles[n]
            if qtypes[n] == GGMLQuantizationType.Q3_K:
                h.act_order = False  # reference gptq.py:204-206
            if h.owner_rank == rank or h._row_split_active():
                if self.verbose:
                    print(f"[rank {rank}] Quantizing {n} with {qtypes[n].name}.")
                h.make_working_copy()
                results[n] = h.compute(qtypes[n])
        # phase 2: exchange, write back the dequantized weight, save
        for n, h in handles.items():
            qweight, d, s, dmin, m = h.exchange(results.get(n), qtypes[n])
            h.layer.weight.data = dequantize_linear_weight(qtypes[n], qweight, d, s, dmin, m,
                                                           out_dtype=h.layer.weight.data.dtype)
            h.reset()
            self._save(n, qtypes[n], qweight, d, s, dmin, m)

    def _quant_non_block_module(self, w: torch.Tensor, q_type: GGMLQuantizationType):
        """RTN for embed / lm_head (reference quantizer.py:278-330)."""
        kw = self.quantizer_kwargs
        # fp16 / bf16 weights: the reference runs make_*quants in the model dtype (quantizer.py:109,195);
        # gq_rtn_quantize reproduces that per-op rounding.  non_block_fp32 opts into an fp32 search instead.
        if w.dtype != torch.float32 and self.non_block_fp32:
            w = w.float()
        return _ops.rtn_quantize(w.contiguous(), int(q_type), kw.get("rmin", -1.0), kw.get("rdelta", 0.1),
                                 kw.get("nstep", 20))

    def _quant_and_save_non_block(self, name, module, quant_config):
        if self.verbose:
            dist_utils.print_on_main(f"Processing {name}.")
        q_type = quant_config.get(name.split(".")[-1], GGMLQuantizationType.Q6_K)  # quantizer.py:107,192
        qweight, d, s, dmin, m = self._quant_non_block_module(module.weight, q_type)
        module.weight.data = dequantize_linear_weight(q_type, qweight, d, s, dmin, m,
                                                      out_dtype=module.weight.data.dtype)
        self._save(name, q_type, qweight, d, s, dmin, m)
