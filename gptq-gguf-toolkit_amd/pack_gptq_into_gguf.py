#!/usr/bin/env python3
"""pack_gptq_into_gguf.py -- HF Llama directory + quantized directory -> .gguf

Keeps the reference CLI (`model --dir_model_quant --outfile --outtype`,
quant/gptq/pack_gptq_into_gguf.py:8792-8884) and reproduces the part of that 9 kLoC llama.cpp
converter fork that touches the hot path:
  * `prepare_tensors` quantized branch (:282-349): directory-name lookup, q_type -> ggml type,
    the q_proj/k_proj row un-permute applied to ALL FIVE tensors (:320-324), dispatch to the
    packers, tensor registered with raw_dtype = q_type and byte shape [R, C/256*type_size];
  * `LlamaModel.permute` / `modify_tensors` (:2177-2183, :2217-2221) and the HF -> GGUF tensor names.
  * the Llama metadata in the reference's order (:412-441, :594-638, :2160-2175), `rope_freqs.weight` for
    rope_type "llama3" (:2259-2287), linear rope-scaling keys, Mixtral's router and stacked expert tensors.
Everything else of the fork (110 other architectures, SentencePiece / Mistral vocabularies, split files, remote
models) is out of scope and REFUSED loudly rather than mis-written.  The container is written by gguf_writer.py (spec-level; whole-file byte
parity with gguf-py is unpinned), the tensor payloads by the GPU bit-packers.
"""
import argparse
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import gptq_gguf_toolkit_amd  # noqa: F401
    from gptq_gguf_toolkit_amd import packing_utils
    from gptq_gguf_toolkit_amd.gguf_writer import GGMLType, GGUFValueType, GGUFWriter
else:
    from . import packing_utils
    from .gguf_writer import GGMLType, GGUFValueType, GGUFWriter

FTYPE = {"f32": (0, GGMLType.F32), "f16": (1, GGMLType.F16), "bf16": (32, GGMLType.BF16)}  # LlamaFileType ids


def permute(weights: torch.Tensor, n_head: int, n_head_kv):
    """Undo HF's rotary row layout (reference :2177-2183); works for any trailing shape."""
    if n_head_kv is not None and n_head != n_head_kv:
        n_head = n_head_kv
    return (weights.reshape(n_head, 2, weights.shape[0] // n_head // 2, *weights.shape[1:])
            .swapaxes(1, 2).reshape(weights.shape))


_BLOCK_TABLE = {"input_layernorm.weight": "attn_norm.weight", "post_attention_layernorm.weight": "ffn_norm.weight",
                "self_attn.q_proj.weight": "attn_q.weight", "self_attn.k_proj.weight": "attn_k.weight",
                "self_attn.v_proj.weight": "attn_v.weight", "self_attn.o_proj.weight": "attn_output.weight",
                "mlp.gate_proj.weight": "ffn_gate.weight", "mlp.up_proj.weight": "ffn_up.weight",
                "mlp.down_proj.weight": "ffn_down.weight",
                # Mixtral (registered under LlamaModel, reference :2120-2124): router + the merged expert tensors
                "block_sparse_moe.gate.weight": "ffn_gate_inp.weight",
                "feed_forward.experts.w1.weight": "ffn_gate_exps.weight",
                "feed_forward.experts.w2.weight": "ffn_down_exps.weight",
                "feed_forward.experts.w3.weight": "ffn_up_exps.weight"}


def map_tensor_name(name: str) -> str:
    """HF Llama / Mixtral -> GGUF tensor names (gguf-py tensor_mapping, llama arch)."""
    if name == "model.embed_tokens.weight":
        return "token_embd.weight"
    if name == "model.norm.weight":
        return "output_norm.weight"
    if name == "lm_head.weight":
        return "output.weight"
    parts = name.split(".")
    if parts[0] == "model":
        parts = parts[1:]
    if len(parts) >= 4 and parts[0] == "layers" and parts[1].isdecimal():
        rest = ".".join(parts[2:])
        if rest in _BLOCK_TABLE:
            return f"blk.{parts[1]}.{_BLOCK_TABLE[rest]}"
    raise ValueError(f"Can not map tensor {name!r}")


def iter_hf_tensors(dir_model: Path):
    from safetensors import safe_open
    files = sorted(dir_model.glob("*.safetensors"))
    if not files:
        raise FileNotFoundError(f"no *.safetensors in {dir_model}")
    for fn in files:
        with safe_open(str(fn), framework="pt", device="cpu") as f:
            for k in f.keys():
                yield k, f.get_tensor(k)


def add_tokenizer(w: GGUFWriter, dir_model: Path, vocab_size: int):
    """The BPE (`gpt2`) vocabulary path of the reference's set_vocab (:2126-2139 falls through sentencepiece and
    llama_hf to _set_vocab_gpt2 for Llama-3).  SentencePiece checkpoints (tokenizer.model: Llama-2, Mistral, Mixtral)
    need the `llama` tokenizer model with scores and byte fallback, which is not reproduced: refused, not mis-written."""
    if (dir_model / "tokenizer.model").exists():
        raise NotImplementedError("SentencePiece vocabulary (tokenizer.model): only the BPE tokenizer.json path of "
                                  "the reference converter is reproduced; pass --no_vocab to write tensors only")
    tj = dir_model / "tokenizer.json"
    if not tj.exists():
        raise FileNotFoundError(f"{tj} not found: no vocabulary to write (pass --no_vocab to write tensors only)")
    tok = json.load(open(tj, encoding="utf-8"))
    if tok["model"].get("type", "BPE") != "BPE" or tok["model"].get("byte_fallback"):
        raise NotImplementedError("tokenizer.json is not a byte-level BPE vocabulary (SentencePiece-derived models need "
                                  "the `llama` tokenizer model, which is not reproduced)")
    vocab = tok["model"]["vocab"]
    added = {a["id"]: a for a in tok.get("added_tokens", [])}
    rev = {i: t for t, i in vocab.items()}
    tokens, types = [], []
    for i in range(vocab_size):
        if i in added:
            tokens.append(added[i]["content"])
            types.append(3 if added[i].get("special") else 4)  # CONTROL / USER_DEFINED
        elif i in rev:
            tokens.append(rev[i])
            types.append(1)  # NORMAL
        else:
            tokens.append(f"[PAD{i}]")
            types.append(5)  # UNUSED
    merges = [m if isinstance(m, str) else " ".join(m) for m in tok["model"].get("merges", [])]
    if vocab_size != 128256:
        print("warning: tokenizer.ggml.pre is written as 'llama-bpe' (the Llama-3 pre-tokenizer); the reference "
              "identifies the pre-tokenizer by a hash of a probe string, which is not reproduced", file=sys.stderr)
    w.add_string("tokenizer.ggml.model", "gpt2")
    w.add_string("tokenizer.ggml.pre", "llama-bpe")
    w.add_array("tokenizer.ggml.tokens", tokens, GGUFValueType.STRING)
    w.add_array("tokenizer.ggml.token_type", types, GGUFValueType.INT32)
    w.add_array("tokenizer.ggml.merges", merges, GGUFValueType.STRING)
    cfgp = dir_model / "tokenizer_config.json"
    cfg = json.load(open(cfgp, encoding="utf-8")) if cfgp.exists() else {}
    tok2id = {t: i for i, t in enumerate(tokens)}
    for key, name in (("bos_token", "bos_token_id"), ("eos_token", "eos_token_id"), ("pad_token", "padding_token_id")):
        v = cfg.get(key)
        v = v.get("content") if isinstance(v, dict) else v
        if v in tok2id:
            w.add_uint32(f"tokenizer.ggml.{name}", tok2id[v])
    if "add_prefix_space" in cfg:  # :2152-2153
        w.add_bool("tokenizer.ggml.add_space_prefix", bool(cfg["add_prefix_space"]))


def size_label(total_params: int, expert_params: int = 0, expert_count: int = 0) -> str:
    """gguf-py's computed `general.size_label` (used when the model name carries none): the parameter count with
    two significant digits ("8.0B", "1.2B"), for MoE "<experts>x<shared + one expert>" ("8x7.2B").
    `expert_params` = parameters of ALL experts together."""
    def rnd(n, min_digits=2):
        for lim, unit in ((1e15, "Q"), (1e12, "T"), (1e9, "B"), (1e6, "M")):
            if n > lim:
                v, u = n / lim, unit
                break
        else:
            v, u = n * 1e-3, "K"
        fix = max(min_digits - len(str(round(v)).lstrip("0")), 0)
        return f"{v:.{fix}f}{u}"
    if expert_count:
        return f"{expert_count}x{rnd(total_params - expert_params + expert_params // expert_count)}"
    return rnd(total_params)


def rope_freqs_llama3(hp: dict):
    """generate_extra_tensors (reference :2259-2287): per-frequency RoPE factors of rope_type "llama3"."""
    import math
    rs = hp["rope_scaling"]
    base = hp.get("rope_theta", 10000.0)
    dim = hp.get("head_dim") or hp["hidden_size"] // hp["num_attention_heads"]
    freqs = 1.0 / (base ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
    factor, low, high = rs.get("factor", 8.0), rs.get("low_freq_factor", 1.0), rs.get("high_freq_factor", 4.0)
    old_ctx = rs.get("original_max_position_embeddings", hp.get("original_max_position_embeddings", 8192))
    low_wl, high_wl = old_ctx / low, old_ctx / high
    out = []
    for freq in freqs:
        wl = 2 * math.pi / freq
        if wl < high_wl:
            out.append(1)
        elif wl > low_wl:
            out.append(factor)
        else:
            smooth = (old_ctx / wl - low) / (high - low)
            out.append(1 / ((1 - smooth) / factor + smooth))
    return torch.tensor(out, dtype=torch.float32)


def convert(dir_model: Path, dir_model_quant: Path, outfile: Path, outtype: str = "f16", verbose: bool = False,
            vocab: bool = True):
    hp = json.load(open(dir_model / "config.json"))
    arch = hp.get("architectures", ["LlamaForCausalLM"])[0]
    if arch not in ("LlamaForCausalLM", "LLaMAForCausalLM", "MistralForCausalLM", "MixtralForCausalLM"):
        raise NotImplementedError(f"Model {arch} is not supported by this packer (Llama family / Mixtral only)")
    n_head = hp["num_attention_heads"]
    n_kv = hp.get("num_key_value_heads", n_head)
    n_experts = hp.get("num_local_experts")
    file_type, out_ggml = FTYPE[outtype]
    rope_scaling = hp.get("rope_scaling") or {}
    rope_type = str(rope_scaling.get("rope_type", rope_scaling.get("type", ""))).lower()
    if rope_type not in ("", "default", "linear", "llama3"):
        raise NotImplementedError(f"rope_scaling type {rope_type!r}: the reference's LlamaModel writes linear scaling "
                                  "keys and llama3 rope_freqs only (:2172-2175, :2259-2287); nothing else is reproduced")

    # ---- tensors first (the reference prepares them before the metadata: write() :444-447), extra tensors lead
    w = GGUFWriter(str(outfile), "llama")
    index_map = {p.name: p for p in dir_model_quant.iterdir() if p.is_dir()} if dir_model_quant else {}  # :285-289
    tied = hp.get("tie_word_embeddings", False)
    names_seen = set()
    total_params = expert_params = 0
    experts = {}  # (bid, wid) -> {expert id: (payload or tensor, q_type or None)}

    def add_plain(new_name, data):
        if data.dtype not in (torch.float16, torch.float32):
            data = data.to(torch.float32)
        # n_dims <= 1, norms and the MoE router stay F32 (:355-376); the rest takes --outtype (:395-410)
        if data.dim() <= 1 or new_name.endswith("_norm.weight") or new_name.endswith("ffn_gate_inp.weight") \
                or not new_name.endswith(".weight") or outtype == "f32":
            w.add_tensor(new_name, data.to(torch.float32).numpy())
        elif outtype == "f16":
            w.add_tensor(new_name, data.to(torch.float16).numpy())
        else:
            bf = data.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
            w.add_tensor(new_name, bf.view(np.uint8).reshape(*bf.shape[:-1], -1), raw_dtype=out_ggml)

    if rope_type == "llama3":
        w.add_tensor("rope_freqs.weight", rope_freqs_llama3(hp).numpy())
    for name, data in iter_hf_tensors(dir_model):
        if name.endswith((".attention.masked_bias", ".attention.bias", ".rotary_emb.inv_freq")):
            continue
        names_seen.add(name)
        total_params += data.numel()
        base = name.removesuffix(".weight")  # :305-306
        is_q, is_k = name.endswith("q_proj.weight"), name.endswith("k_proj.weight")
        payload = q_type = None
        if base in index_map:
            qd = torch.load(str(index_map[base] / "data.pth"), map_location="cpu", weights_only=True)
            q_type = int(qd["q_type"])
            five = [qd["qweight"], qd["super_group_scale"], qd["group_scale_quant"], qd["super_group_zero"],
                    qd["group_zero_quant"]]
            if is_q:
                five = [permute(t, n_head, n_head) for t in five]      # :320-324 via modify_tensors
            elif is_k:
                five = [permute(t, n_head, n_kv) for t in five]
            payload = packing_utils.pack_tensor(q_type, *five)          # :326-336
        elif is_q:
            data = permute(data, n_head, n_head)
        elif is_k:
            data = permute(data, n_head, n_kv)
        if ".block_sparse_moe.experts." in name:
            # :2223-2255 merges the experts of a block into one 3-D tensor per w1/w2/w3.  The reference's quantized
            # branch indexes modify_tensors(...)[0], which is empty for all but a block's last expert tensor -- it
            # has no working behaviour for quantized experts.  Here: the packed payloads of the experts are stacked
            # [n_experts, R, C/256*type_size] (ggml ne = [C, R, n_experts]); all experts of a tensor share a type.
            parts = name.split(".")
            bid, xid, wid = int(parts[2]), int(parts[5]), parts[6]
            expert_params += data.numel()
            slot = experts.setdefault((bid, wid), {})
            slot[xid] = (payload if payload is not None else data, q_type)
            if len(slot) == n_experts:
                types = {t for _, t in slot.values()}
                if len(types) != 1:
                    raise ValueError(f"experts of layers.{bid} {wid} carry different types {sorted(map(str, types))}")
                new_name = map_tensor_name(f"layers.{bid}.feed_forward.experts.{wid}.weight")
                qt = types.pop()
                if qt is None:
                    add_plain(new_name, torch.stack([slot[e][0] for e in range(n_experts)], dim=0))
                else:
                    stacked = np.stack([slot[e][0] for e in range(n_experts)], axis=0)
                    if verbose:
                        print(f"{new_name:28s} {n_experts} x {tuple(data.shape)} --> ggml type {qt}, {stacked.nbytes} bytes")
                    w.add_tensor(new_name, stacked, raw_dtype=qt)
                del experts[(bid, wid)]
            continue
        new_name = map_tensor_name(name)
        if payload is not None:
            if verbose:
                print(f"{new_name:28s} {tuple(data.shape)} --> ggml type {q_type}, {payload.nbytes} bytes")
            w.add_tensor(new_name, payload, raw_dtype=q_type)           # :344-348
        else:
            add_plain(new_name, data)
    if experts:
        raise ValueError(f"Unprocessed experts: {sorted(experts)}")   # :2289-2296
    if "lm_head.weight" not in names_seen and not tied:
        print("warning: no lm_head.weight in the checkpoint and tie_word_embeddings is false", file=sys.stderr)

    # ---- metadata in the reference's order: prepare_metadata (:412-441) -> set_gguf_parameters (:594-638,
    # :2160-2175) -> quantization version -> set_vocab (:590-592)
    w.add_string("general.type", "model")
    w.add_string("general.name", hp.get("_name_or_path") or dir_model.name)
    if total_params > 0:
        w.add_string("general.size_label", size_label(total_params, expert_params, n_experts or 0))
    w.add_uint32("llama.block_count", hp["num_hidden_layers"])
    for key, gg, kind in (("max_position_embeddings", "llama.context_length", "u"), ("hidden_size", "llama.embedding_length", "u"),
                          ("intermediate_size", "llama.feed_forward_length", "u"), ("num_attention_heads", "llama.attention.head_count", "u"),
                          ("num_key_value_heads", "llama.attention.head_count_kv", "u"), ("rope_theta", "llama.rope.freq_base", "f"),
                          ("rms_norm_eps", "llama.attention.layer_norm_rms_epsilon", "f"),
                          ("num_local_experts", "llama.expert_count", "u"), ("num_experts_per_tok", "llama.expert_used_count", "u")):
        if hp.get(key) is not None:
            (w.add_uint32 if kind == "u" else w.add_float32)(gg, hp[key])
    if hp.get("head_dim") is not None:
        w.add_uint32("llama.attention.key_length", hp["head_dim"])
        w.add_uint32("llama.attention.value_length", hp["head_dim"])
    w.add_uint32("general.file_type", file_type)
    w.add_uint32("llama.vocab_size", hp["vocab_size"])
    w.add_uint32("llama.rope.dimension_count", hp.get("head_dim") or hp["hidden_size"] // n_head)
    if rope_type == "linear" and "factor" in rope_scaling:
        w.add_string("llama.rope.scaling.type", "linear")
        w.add_float32("llama.rope.scaling.factor", rope_scaling["factor"])
    w.add_uint32("general.quantization_version", 2)
    if vocab:
        add_tokenizer(w, dir_model, hp["vocab_size"])
    w.write()
    return outfile


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Convert a HF Llama model + GPTQ K-quant results to a GGUF file")
    p.add_argument("model", type=Path, help="directory containing the original HF model")
    p.add_argument("--dir_model_quant", type=Path, required=True, help="directory written by quant.py (--save_dir)")
    p.add_argument("--outfile", type=Path, required=True)
    p.add_argument("--outtype", type=str, choices=["f32", "f16", "bf16"], default="f16",
                   help="type of the tensors that were NOT quantized")
    p.add_argument("--verbose", action="store_true")
    p.add_argument("--no_vocab", action="store_true", help="beyond the reference: write tensors and model metadata only")
    return p.parse_args(argv)


def main(argv=None):
    a = parse_args(argv)
    out = convert(a.model, a.dir_model_quant, a.outfile, a.outtype, a.verbose, vocab=not a.no_vocab)
    print(f"Model successfully exported to {out}")


if __name__ == "__main__":
    main()
