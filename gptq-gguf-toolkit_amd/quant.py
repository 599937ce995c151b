#!/usr/bin/env python3
"""CLI entry: same flags, defaults and timing scope as the reference's quant/gptq/quant.py
(:18-142 flags, :177-179 calibration sharding, :183-217 quant_config, :251-254 timing).

One process per GPU (torchrun), `torch.distributed` backend "nccl" == RCCL over xGMI on ROCm.
"""
import argparse
import json
import os
import sys
import time

# one hardware queue per HIP stream: block_schedule.BlockSchedule runs the input groups of a block as concurrent
# chains on 4 streams, and the runtime's default of 4 queues makes two of them share one
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import torch
import torch.distributed as dist

if __package__ in (None, ""):  # run as a script: make the package importable under its alias
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import gptq_gguf_toolkit_amd  # noqa: F401
    from gptq_gguf_toolkit_amd import dist_utils
    from gptq_gguf_toolkit_amd.quant_utils import GGMLQuantizationType
    from gptq_gguf_toolkit_amd.quantizer import Quantizer
else:
    from . import dist_utils
    from .quant_utils import GGMLQuantizationType
    from .quantizer import Quantizer

Q_NAMES = ["Q2_K", "Q3_K", "Q4_K", "Q5_K", "Q6_K"]
DEFAULT_KEYS = ["q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "down_proj", "up_proj", "embed_tokens", "lm_head"]


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    # Model params
    p.add_argument("--model_name_or_path", type=str, required=True, help="The name or path to quantized model.")
    p.add_argument("--tokenizer_name", type=str, default=None)
    p.add_argument("--quantizable_modules", type=str, required=True, help="Regex for modules to quantize")
    p.add_argument("--pre_block_modules", nargs="+", type=str, required=True)
    p.add_argument("--block_modules", type=str, required=True)
    p.add_argument("--post_block_modules", nargs="+", type=str, default=[])
    p.add_argument("--quant_non_block_modules", action="store_true")
    # Data params
    p.add_argument("--calibration_data", type=str, required=True)
    p.add_argument("--calibration_tokens", default=int(2 ** 20), type=int)
    p.add_argument("--calibration_sequence_length", default=None, type=int)
    # Quantization params
    p.add_argument("--quant_scale", type=str, default="absmax", choices=["absmax", "mse"])
    p.add_argument("--act_order", action="store_true")
    p.add_argument("--static_groups", action="store_true")
    p.add_argument("--rel_damp", type=float, default=1e-2)
    p.add_argument("--block_size", type=int, default=128)
    p.add_argument("--default_bit_width", type=str, default="Q4_K")
    p.add_argument("--bit_width_configuration", type=str, default=None)
    p.add_argument("--rmin", type=float, default=-1.0)
    p.add_argument("--rdelta", type=float, default=0.1)
    p.add_argument("--nstep", type=int, default=20)
    p.add_argument("--log_wandb", default=False, action="store_true")
    # Misc params
    p.add_argument("--dtype", type=str, default="auto", choices=["auto", "float16", "float32", "bfloat16"])
    p.add_argument("--seed", default=0, type=int)
    p.add_argument("--low_cpu_mem_usage", action="store_true")
    p.add_argument("--attn_implementation", type=str, default=None, choices=["eager", "sdpa", "flash_attention_2"])
    p.add_argument("--cpu_offload_modules", action="store_true")
    p.add_argument("--cpu_offload_activations", action="store_true")
    p.add_argument("--eval_perplexity", action="store_true")
    p.add_argument("--eval_sequence_length", type=int, default=4096)
    p.add_argument("--verbose", action="store_true")
    p.add_argument("--save_dir", type=str, required=True)
    # beyond the reference
    p.add_argument("--calibration_batch", type=int, default=1,
                   help="calibration samples per block forward (the reference runs 1; the Hessians are the same sums)")
    p.add_argument("--non_block_fp32", action="store_true",
                   help="run the embed/lm_head scale search in fp32 (the reference runs it in the model dtype)")
    p.add_argument("--full_forward1", action="store_true",
                   help="beyond the reference: by default forward #1 of a block stops at the last hooked Linear (its output is "
                        "discarded, quantizer.py:150-151) and forward #2 of the last block is not run; this flag runs both in full "
                        "like the reference -- same saved bytes either way")
    p.add_argument("--fused_forward", nargs="?", const="all", default="exact", choices=["off", "exact", "all"],
                   help="HIP kernels for the elementwise modules of the calibration forward (the reference runs the HF "
                        "eager modules): exact (default) = rotary embedding, SwiGLU and RMSNorm, bit-identical to HF eager "
                        "(RMSNorm checked at run time); all (or the bare flag) = also the free-order RMSNorm kernel, "
                        "<= 2 ulp, where the exact one is not verified; off = none")
    return p.parse_args(argv)


def build_quant_config(default_bit_width, bit_width_configuration):
    """reference quant.py:183-217 (the JSON file, when given, REPLACES the default map)."""
    if default_bit_width is None and bit_width_configuration is None:
        raise ValueError("Either default_bit_width or bit_width_configuration must be provided.")
    quant_config = {}
    if default_bit_width is not None:
        if default_bit_width not in Q_NAMES:
            raise ValueError("default_bit_width must be one of [Q2_K, Q3_K, Q4_K, Q5_K, Q6_K]")
        quant_config = {k: GGMLQuantizationType[default_bit_width] for k in DEFAULT_KEYS}
    if bit_width_configuration is not None:
        if not os.path.isfile(bit_width_configuration):
            raise ValueError("bit_width_configuration must be a valid file path.")
        with open(bit_width_configuration) as f:
            cfg = json.load(f)
        quant_config = {}
        for key, value in cfg.items():
            if value not in Q_NAMES:
                raise ValueError("All bit widths in bit_width_configuration must be one of "
                                 "[Q2_K, Q3_K, Q4_K, Q5_K, Q6_K]")
            quant_config[key] = GGMLQuantizationType[value]
    return quant_config


def load_calibration(path_or_name, num_tokens, seq_len, tokenizer):
    """The reference's `.pt` branch (data_utils.py:134-136): a list of [1, L] id tensors.  The
    network datasets (wikitext2 / c4 / fineweb_edu) are out of scope here."""
    if os.path.isfile(path_or_name):
        data = torch.load(path_or_name)[: num_tokens // seq_len]
        return [s[:, :seq_len] for s in data]
    raise ValueError(f"calibration_data must be a .pt file of token-id tensors (got {path_or_name!r}); "
                     "dataset downloads are not part of this package")


def main(argv=None):
    args = parse_args(argv)
    if args.eval_perplexity:  # refused BEFORE any work (the reference evaluates WikiText-2 here: a dataset download)
        raise SystemExit("--eval_perplexity is not available in this package (WikiText-2 needs a dataset download); "
                         "run the quantization without it and evaluate the saved model separately")
    if dist.is_available() and "RANK" in os.environ:
        dist.init_process_group(backend="nccl", init_method="env://")  # RCCL
    try:
        _run(args)
    finally:  # also when the run raises: torchrun then reports the real error, not a hung rendezvous
        if dist_utils.is_dist_available_and_initialized():
            dist.destroy_process_group()


def _run(args):
    from transformers import AutoModelForCausalLM, AutoTokenizer
    world_size, rank = dist_utils.get_world_size(), dist_utils.get_rank()
    device = f"cuda:{int(os.environ.get('LOCAL_RANK', rank))}"
    torch.cuda.set_device(device)

    import transformers
    dtype_kw = "dtype" if int(transformers.__version__.split(".")[0]) >= 5 else "torch_dtype"  # renamed in 5.x
    model = AutoModelForCausalLM.from_pretrained(
        args.model_name_or_path, trust_remote_code=True, **{dtype_kw: args.dtype},
        low_cpu_mem_usage=args.low_cpu_mem_usage, attn_implementation=args.attn_implementation)
    if not args.cpu_offload_modules:
        model = model.to(device)
    tokenizer = None
    if not os.path.isfile(args.calibration_data) or args.eval_perplexity:
        tokenizer = AutoTokenizer.from_pretrained(args.tokenizer_name or args.model_name_or_path, use_fast=False)
    args.calibration_sequence_length = args.calibration_sequence_length or model.config.max_position_embeddings
    data = load_calibration(args.calibration_data, args.calibration_tokens, args.calibration_sequence_length, tokenizer)
    if world_size > 1:
        data = dist_utils.shard_calibration(data, rank, world_size)
    data = [([], {"input_ids": ids}) for ids in data]
    dist_utils.barrier()

    quant_config = build_quant_config(args.default_bit_width, args.bit_width_configuration)
    quantizer = Quantizer(
        model, data_loader=data, quantizable_modules=args.quantizable_modules,
        quantizer_kwargs=dict(rel_damp=args.rel_damp, block_size=args.block_size, act_order=args.act_order,
                              quant_scale=args.quant_scale, static_groups=args.static_groups, rmin=args.rmin,
                              rdelta=args.rdelta, nstep=args.nstep, verbose=args.verbose),
        pre_block_modules=args.pre_block_modules, block_modules=args.block_modules,
        post_block_modules=args.post_block_modules, quant_non_block_modules=args.quant_non_block_modules,
        cpu_offload_modules=args.cpu_offload_modules, cpu_offload_activations=args.cpu_offload_activations,
        device=device, verbose=args.verbose, save_dir=args.save_dir, non_block_fp32=args.non_block_fp32,
        calibration_batch=args.calibration_batch, fused_forward=args.fused_forward,
        interrupt_forward1=not args.full_forward1)
    if dist_utils.is_main():
        os.makedirs(args.save_dir, exist_ok=True)
    dist_utils.barrier()

    t1 = time.perf_counter()
    quantizer.quantize(quant_config)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    dist_utils.print_on_main(f"Quantization took {(t2 - t1)} s.")
    dist_utils.barrier()


if __name__ == "__main__":
    main()
