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
  * `LlamaModel.set_vocab` (:2126-2139): the SentencePiece vocabulary (`tokenizer.model` -> tokenizer model "llama",
    scores, token types, byte fallback; :1018-1118 -- TinyLlama, Llama-2, Mistral, Mixtral) and the byte-level BPE one
    (Llama-3), both closed by gguf.SpecialVocab's keys (special token ids, add_bos/eos flags, chat template).
Everything else of the fork (110 other architectures, Mistral-format / tekken vocabularies, split files, remote
models) is out of scope and REFUSED loudly rather than mis-written.  The container is written by gguf_writer.py
(spec-level; whole-file byte identity with gguf-py 0.17.1 is UNPINNED: that package cannot be installed here and no
reference-produced file exists to compare with), the tensor payloads by the GPU bit-packers.
"""
import argparse
import json
import os
import sys
from pathlib import Path
from typing import Optional

import numpy as np
import torch

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import gptq_gguf_toolkit_amd  # noqa: F401
    from gptq_gguf_toolkit_amd import packing_utils
    from gptq_gguf_toolkit_amd.gguf_writer import GGMLType, GGUFValueType, GGUFWriter, QuantError, quantize_q8_0
else:
    from . import packing_utils
    from .gguf_writer import GGMLType, GGUFValueType, GGUFWriter, QuantError, quantize_q8_0

# --outtype -> (LlamaFileType id, ggml type of the tensors GPTQ did not quantize); reference :8956-8964
FTYPE = {"f32": (0, GGMLType.F32), "f16": (1, GGMLType.F16), "bf16": (32, GGMLType.BF16), "q8_0": (7, GGMLType.Q8_0)}
OUTTYPES = ["f32", "f16", "bf16", "q8_0", "tq1_0", "tq2_0", "auto"]  # the reference's choices (:8804)


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
    for k, _, get in iter_hf_entries(dir_model):
        yield k, get()


def iter_hf_entries(dir_model: Path):
    """-> (name, shape, get) in the checkpoint's order; get() reads the tensor.  A tensor GPTQ replaced is never read: the
    converter needs its name and shape only (16 GB of an 8B checkpoint otherwise, for nothing)."""
    from safetensors import safe_open
    files = sorted(dir_model.glob("*.safetensors"))
    if not files:
        raise FileNotFoundError(f"no *.safetensors in {dir_model}")
    for fn in files:
        with safe_open(str(fn), framework="pt", device="cpu") as f:
            for k in f.keys():
                yield k, tuple(f.get_slice(k).get_shape()), (lambda f=f, k=k: f.get_tensor(k))


# llama.cpp token types (reference pack_gptq_into_gguf.py:47-53)
TOK_NORMAL, TOK_UNKNOWN, TOK_CONTROL, TOK_USER_DEFINED, TOK_UNUSED, TOK_BYTE = 1, 2, 3, 4, 5, 6
_SPECIAL_TYPES = ("bos", "eos", "unk", "sep", "pad", "cls", "mask")
# the GGUF key of each special token id (gguf-py constants.py Keys.Tokenizer; "seperator" is the spec's spelling)
_SPECIAL_KEYS = {"bos": "bos_token_id", "eos": "eos_token_id", "unk": "unknown_token_id", "sep": "seperator_token_id",
                 "pad": "padding_token_id", "cls": "cls_token_id", "mask": "mask_token_id"}


def does_token_look_special(token: str) -> bool:
    """reference pack_gptq_into_gguf.py:649-670: added tokens that ought to be control tokens whatever their flag."""
    return (token in ("<pad>", "<mask>", "<2mass>", "[@BOS@]")
            or (token.startswith("<|") and token.endswith("|>"))
            or (token.startswith("<\uff5c") and token.endswith("\uff5c>"))
            or (token.startswith("<unused") and token.endswith(">")))


def add_special_vocab(w: GGUFWriter, dir_model: Path, n_vocab: int, merges=None) -> None:
    """What `gguf.SpecialVocab(dir_model, n_vocab=...).add_to_gguf(writer)` leaves in the file (the reference's vocab
    paths all end with it, pack_gptq_into_gguf.py:1027-1028): merges (BPE only), the special token ids, the
    add_<type>_token flags, the chat template.  gguf-py 0.17.1 is not installable here; this follows its published
    behaviour: for every type in (bos, eos, unk, sep, pad, cls, mask) tokenizer_config.json's "<type>_token" (a string or
    {"content": ...}) is looked up among tokenizer.json's added_tokens, then config.json's "<type>_token_id" fills
    what is still unset; ids >= n_vocab are dropped; "add_<type>_token" booleans and "chat_template" come from
    tokenizer_config.json (chat_template.json as the alternative source)."""
    ids, add_flags, chat_template = {}, {}, None

    def set_id(typ, tid):
        if not isinstance(tid, int) or isinstance(tid, bool) or tid < 0 or typ in ids:
            return
        if n_vocab is None or tid < n_vocab:
            ids[typ] = tid

    tj, tcj, cj = dir_model / "tokenizer.json", dir_model / "tokenizer_config.json", dir_model / "config.json"
    added = json.load(open(tj, encoding="utf-8")).get("added_tokens", []) if tj.is_file() else []
    if tcj.is_file():
        cfg = json.load(open(tcj, encoding="utf-8"))
        alt = dir_model / "chat_template.json"
        tmpl = cfg.get("chat_template", json.load(open(alt, encoding="utf-8")).get("chat_template") if alt.is_file() else None)
        if isinstance(tmpl, (str, list)):
            chat_template = tmpl
        for typ in _SPECIAL_TYPES:
            if isinstance(cfg.get(f"add_{typ}_token"), bool):
                add_flags[typ] = cfg[f"add_{typ}_token"]
            entry = cfg.get(f"{typ}_token")
            content = entry if isinstance(entry, str) else (entry.get("content") if isinstance(entry, dict) else None)
            if isinstance(content, str):
                set_id(typ, next((a.get("id") for a in added if a.get("content") == content), None))
    if cj.is_file():
        cfg = json.load(open(cj, encoding="utf-8"))
        for typ in _SPECIAL_TYPES:
            set_id(typ, cfg.get(f"{typ}_token_id"))
    if merges:
        w.add_array("tokenizer.ggml.merges", merges, GGUFValueType.STRING)
    for typ, tid in ids.items():
        w.add_uint32(f"tokenizer.ggml.{_SPECIAL_KEYS[typ]}", tid)
    for typ, flag in add_flags.items():
        w.add_bool(f"tokenizer.ggml.add_{typ}_token", flag)
    if isinstance(chat_template, list):  # several named templates: "default" is the one llama.cpp reads first
        named = {t.get("name"): t.get("template") for t in chat_template if isinstance(t, dict)}
        chat_template = named.get("default")
        for name, text in named.items():
            if name != "default" and isinstance(text, str):
                w.add_string(f"tokenizer.chat_template.{name}", text)
    if isinstance(chat_template, str):
        w.add_string("tokenizer.chat_template", chat_template)


def create_vocab_sentencepiece(dir_model: Path, vocab_size: Optional[int]):
    """reference pack_gptq_into_gguf.py:1030-1118 (_create_vocab_sentencepiece): (tokens, scores, token types) of a
    SentencePiece checkpoint -- every piece with its score and its type (normal / unknown / control / unused / byte),
    added_tokens.json entries as USER_DEFINED with score -1000, tokenizer_config.json's added_tokens_decoder entries as
    CONTROL (flagged or special-looking) or USER_DEFINED (with the U+2581 pre-normalisation), [PAD<i>] / UNUSED for
    ids the model file does not cover."""
    from sentencepiece import SentencePieceProcessor
    path = dir_model / "tokenizer.model"
    if not path.is_file():
        raise FileNotFoundError(f"File not found: {path}")
    sp = SentencePieceProcessor()
    sp.LoadFromFile(str(path))
    vocab_size = vocab_size or sp.vocab_size()
    tokens = [f"[PAD{i}]" for i in range(vocab_size)]
    scores = [-10000.0] * vocab_size
    types = [TOK_UNUSED] * vocab_size
    for tid in range(min(sp.vocab_size(), vocab_size)):
        tokens[tid] = sp.IdToPiece(tid)
        scores[tid] = float(sp.GetScore(tid))
        types[tid] = (TOK_UNKNOWN if sp.IsUnknown(tid) else TOK_CONTROL if sp.IsControl(tid) else
                      TOK_UNUSED if sp.IsUnused(tid) else TOK_BYTE if sp.IsByte(tid) else TOK_NORMAL)
    added_file = dir_model / "added_tokens.json"
    if added_file.is_file():
        for key, tid in json.load(open(added_file, encoding="utf-8")).items():
            if tid < vocab_size:
                tokens[tid], scores[tid], types[tid] = key, -1000.0, TOK_USER_DEFINED
    tcj = dir_model / "tokenizer_config.json"
    if tcj.is_file():
        for tid, data in json.load(open(tcj, encoding="utf-8")).get("added_tokens_decoder", {}).items():
            tid, token = int(tid), data["content"]
            if tid >= vocab_size:
                continue
            if data.get("special") or does_token_look_special(token):
                types[tid] = TOK_CONTROL
            else:
                token = token.replace("\u2581", " ")  # pre-normalize user-defined spaces (:1104)
                types[tid] = TOK_USER_DEFINED
            scores[tid] = -1000.0
            tokens[tid] = token
    return tokens, scores, types


# True: drop the CodeLlama prefix / suffix / middle token ids as gguf-py 0.17.1's GGUFWriter is believed to (see _add_space_prefix)
GGUF_PY_FIM_COMPAT = False


def add_tokenizer(w: GGUFWriter, dir_model: Path, vocab_size: int):
    """LlamaModel.set_vocab (reference :2126-2139): the SentencePiece vocabulary when the checkpoint has a
    tokenizer.model (Llama-2, TinyLlama, Mistral, Mixtral: tokenizer model "llama", pre-tokenizer "default", scores,
    byte fallback; :1018-1028), else the byte-level BPE path (`gpt2`, Llama-3; :2137).  The middle fallback
    (`_set_vocab_llama_hf`: a tokenizer.json of a SentencePiece-derived vocabulary without tokenizer.model) is not
    reproduced and refused."""
    if (dir_model / "tokenizer.model").is_file():
        tokens, scores, types = create_vocab_sentencepiece(dir_model, vocab_size)
        w.add_string("tokenizer.ggml.model", "llama")
        w.add_string("tokenizer.ggml.pre", "default")
        w.add_array("tokenizer.ggml.tokens", tokens, GGUFValueType.STRING)
        w.add_array("tokenizer.ggml.scores", scores, GGUFValueType.FLOAT32)
        w.add_array("tokenizer.ggml.token_type", types, GGUFValueType.INT32)
        add_special_vocab(w, dir_model, len(tokens))
        _add_space_prefix(w, dir_model, vocab_size)
        return
    tj = dir_model / "tokenizer.json"
    if not tj.exists():
        raise FileNotFoundError(f"{tj} not found: no vocabulary to write (pass --no_vocab to write tensors only)")
    tok = json.load(open(tj, encoding="utf-8"))
    if tok["model"].get("type", "BPE") != "BPE" or tok["model"].get("byte_fallback"):
        raise NotImplementedError("tokenizer.json holds a SentencePiece-derived vocabulary but there is no tokenizer.model "
                                  "next to it (the reference's _set_vocab_llama_hf path is not reproduced)")
    vocab = tok["model"]["vocab"]
    added = {a["id"]: a for a in tok.get("added_tokens", [])}
    rev = {i: t for t, i in vocab.items()}
    tokens, types = [], []
    for i in range(vocab_size):
        if i in added:
            tokens.append(added[i]["content"])
            types.append(TOK_CONTROL if added[i].get("special") or does_token_look_special(added[i]["content"])
                         else TOK_USER_DEFINED)
        elif i in rev:
            tokens.append(rev[i])
            types.append(TOK_NORMAL)
        else:
            tokens.append(f"[PAD{i}]")
            types.append(TOK_UNUSED)
    # merges as [a, b] pairs (transformers >= 4.45): gguf-py's SpecialVocab joins them with a space after encoding the
    # spaces INSIDE a part as chr(ord(' ') + 256), the byte-level alphabet's letter for 0x20
    merges = [m if isinstance(m, str) else " ".join("".join(chr(ord(c) + 256) if c == " " else c for c in part) for part in m)
              for m in tok["model"].get("merges", [])]
    if vocab_size != 128256:
        print("warning: tokenizer.ggml.pre is written as 'llama-bpe' (the Llama-3 pre-tokenizer); the reference "
              "identifies the pre-tokenizer by a hash of a probe string, which is not reproduced", file=sys.stderr)
    w.add_string("tokenizer.ggml.model", "gpt2")
    w.add_string("tokenizer.ggml.pre", "llama-bpe")
    w.add_array("tokenizer.ggml.tokens", tokens, GGUFValueType.STRING)
    w.add_array("tokenizer.ggml.token_type", types, GGUFValueType.INT32)
    add_special_vocab(w, dir_model, len(tokens), merges=merges)
    _add_space_prefix(w, dir_model, vocab_size)


def _add_space_prefix(w: GGUFWriter, dir_model: Path, vocab_size: Optional[int] = None) -> None:
    """The tail of LlamaModel.set_vocab (:2138-2158), in its order."""
    if vocab_size == 32016:
        # CodeLlama only (:2138-2148): a second SpecialVocab(load_merges=False, special_token_types=prefix/suffix/middle/eot)
        # with the four fill-in-the-middle ids set by hand -- written, all four, as the reference's lines ask for.
        # What a particular gguf-py release makes of them cannot be confirmed here (no gguf-py): the writer of
        # 0.17.1 is believed to have add_eot_token_id only and to skip the other three with a warning; GGUF_PY_FIM_COMPAT = True
        # reproduces that guess, the default keeps the keys (no output key is dropped on an unverified premise).
        fim = (("prefix", 32007), ("suffix", 32008), ("middle", 32009), ("eot", 32010))
        for typ, tid in fim:
            if GGUF_PY_FIM_COMPAT and typ != "eot":
                print(f"warning: No handler for special token type {typ} with id {tid} - skipping", file=sys.stderr)
                continue
            w.add_uint32(f"tokenizer.ggml.{typ}_token_id", tid)
        # The same second SpecialVocab reads tokenizer_config.json again; its chat template is already in the file (the first
        # SpecialVocab wrote it).  A duplicate key would make the writer raise: the template is NOT added a second time -- a
        # CodeLlama-Instruct checkpoint converts (a suspected upstream crash is not emulated).
        tcj = dir_model / "tokenizer_config.json"
        tmpl = json.load(open(tcj, encoding="utf-8")).get("chat_template") if tcj.is_file() else None
        if isinstance(tmpl, str) and not any(k == "tokenizer.chat_template" for k, *_ in w.kv):
            w.add_string("tokenizer.chat_template", tmpl)
    cfgp = dir_model / "tokenizer_config.json"
    cfg = json.load(open(cfgp, encoding="utf-8")) if cfgp.exists() else {}
    if vocab_size == 49152:
        # granite small models only (:2156-2158).  (Placed after add_prefix_space in the reference; a checkpoint whose
        # tokenizer_config.json already carries add_bos_token makes gguf-py raise "Duplicated key name" there -- and here.)
        post = [("tokenizer.ggml.add_bos_token", False)]
    else:
        post = []
    if "add_prefix_space" in cfg:  # :2150-2153
        w.add_bool("tokenizer.ggml.add_space_prefix", bool(cfg["add_prefix_space"]))
    for k, v in post:
        w.add_bool(k, v)


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
            vocab: bool = True, pipelined: bool = True, timing: Optional[dict] = None):
    """pipelined (default): a quantized Linear's five tensors are mapped from data.pth (torch.load(mmap=True): no read), and its
    payload is made WHEN THE FILE IS WRITTEN, on a producer thread a few tensors ahead of the file writes -- one upload, the q / k
    row un-permute and gq_pack on the GPU, one download (GGUFWriter.add_tensor_lazy) -- instead of five uploads, a CPU permute
    and ~5 GB of payloads held until write() (pipelined=False: that flow, the reference's :282-349 order of operations).  Same
    file bytes either way (tests/test_host_logic_cpu.py::test_pack_into_gguf...).  `timing` receives seconds per stage."""
    import threading
    import time
    tm = timing if timing is not None else {}
    tm_lock = threading.Lock()

    def clock(key, t0):  # (producers run on several threads: their seconds add up per stage, wall time is the caller's)
        dt = time.perf_counter() - t0
        with tm_lock:
            tm[key] = tm.get(key, 0.0) + dt

    hp = json.load(open(dir_model / "config.json"))
    arch = hp.get("architectures", ["LlamaForCausalLM"])[0]
    if arch not in ("LlamaForCausalLM", "LLaMAForCausalLM", "MistralForCausalLM", "MixtralForCausalLM"):
        raise NotImplementedError(f"Model {arch} is not supported by this packer (Llama family / Mixtral only)")
    n_head = hp["num_attention_heads"]
    n_kv = hp.get("num_key_value_heads", n_head)
    n_experts = hp.get("num_local_experts")
    if outtype in ("tq1_0", "tq2_0"):
        raise NotImplementedError(f"--outtype {outtype}: the ternary encoders of gguf-py (gguf.quants TQ1_0 / TQ2_0) are not "
                                  "reproduced; GPTQ K-quant results are not ternary models")
    if outtype == "auto":
        # LlamaFileType.GUESSED (:144-152): the highest-fidelity 16-bit type for the FIRST tensor of the checkpoint
        _, first = next(iter_hf_tensors(dir_model))
        outtype = "f16" if first.dtype == torch.float16 else "bf16"
        if verbose:
            print(f"choosing --outtype {outtype} from first tensor type ({first.dtype})")
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
        elif outtype == "q8_0":
            try:
                w.add_tensor(new_name, quantize_q8_0(data.numpy()), raw_dtype=GGMLType.Q8_0)
            except QuantError as e:  # :419-424: a row length that is no multiple of 32 falls back to F16
                print(f"{e}, falling back to F16", file=sys.stderr)
                w.add_tensor(new_name, data.to(torch.float16).numpy())
        else:
            bf = data.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
            w.add_tensor(new_name, bf.view(np.uint8).reshape(*bf.shape[:-1], -1), raw_dtype=out_ggml)

    if rope_type == "llama3":
        w.add_tensor("rope_freqs.weight", rope_freqs_llama3(hp).numpy())
    FIVE = ("qweight", "super_group_scale", "group_scale_quant", "super_group_zero", "group_zero_quant")
    for name, shape, get in iter_hf_entries(dir_model):
        if name.endswith((".attention.masked_bias", ".attention.bias", ".rotary_emb.inv_freq")):
            continue
        names_seen.add(name)
        numel = int(np.prod(shape)) if len(shape) else 1
        total_params += numel
        base = name.removesuffix(".weight")  # :305-306
        is_q, is_k = name.endswith("q_proj.weight"), name.endswith("k_proj.weight")
        is_expert = ".block_sparse_moe.experts." in name
        payload = q_type = data = None
        if base in index_map and pipelined and not is_expert:
            t0 = time.perf_counter()
            qd = torch.load(str(index_map[base] / "data.pth"), map_location="cpu", weights_only=True, mmap=True)
            clock("load", t0)
            q_type = int(qd["q_type"])

            def producer(qd=qd, q_type=q_type, heads=(n_head, n_head if is_q else n_kv) if (is_q or is_k) else None):
                t0 = time.perf_counter()
                five = [packing_utils._dev(qd[k]) if torch.is_tensor(qd.get(k)) else None for k in FIVE]   # the one upload (pageable, synchronous)
                clock("h2d", t0)
                t0 = time.perf_counter()
                if heads is not None:
                    five = [permute(t, *heads) if t is not None else None for t in five]      # :320-324 via modify_tensors
                out = packing_utils.pack_tensor_on_device(q_type, *five)                    # :326-336
                if out.is_cuda:
                    torch.cuda.synchronize()
                clock("permute_pack", t0)
                t0 = time.perf_counter()
                host = out.cpu().numpy()
                clock("d2h", t0)
                return host

            new_name = map_tensor_name(name)
            if verbose:
                print(f"{new_name:28s} {tuple(shape)} --> ggml type {q_type}")
            w.add_tensor_lazy(new_name, tuple(qd["qweight"].shape), q_type, producer)       # :344-348
            continue
        t0 = time.perf_counter()
        data = get()
        clock("hf_read", t0)
        if base in index_map:
            t0 = time.perf_counter()
            qd = torch.load(str(index_map[base] / "data.pth"), map_location="cpu", weights_only=True)
            clock("load", t0)
            q_type = int(qd["q_type"])
            five = [qd[k] for k in FIVE]
            t0 = time.perf_counter()
            if is_q:
                five = [permute(t, n_head, n_head) for t in five]      # :320-324 via modify_tensors
            elif is_k:
                five = [permute(t, n_head, n_kv) for t in five]
            clock("permute_cpu", t0)
            t0 = time.perf_counter()
            payload = packing_utils.pack_tensor(q_type, *five)          # :326-336
            clock("h2d_pack_d2h", t0)
        elif is_q:
            data = permute(data, n_head, n_head)
        elif is_k:
            data = permute(data, n_head, n_kv)
        if is_expert:
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
            t0 = time.perf_counter()
            add_plain(new_name, data)
            clock("plain", t0)
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
    t0 = time.perf_counter()
    w.write(tm)
    clock("write_call", t0)
    return outfile


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Convert a HF Llama model + GPTQ K-quant results to a GGUF file")
    p.add_argument("model", type=Path, help="directory containing the original HF model")
    p.add_argument("--dir_model_quant", type=Path, required=True, help="directory written by quant.py (--save_dir)")
    p.add_argument("--outfile", type=Path, required=True)
    p.add_argument("--outtype", type=str, choices=OUTTYPES, default="f16",
                   help="type of the tensors that were NOT quantized: f32 / f16 / bf16 / q8_0, auto = f16 or bf16 after the "
                        "checkpoint's first tensor; tq1_0 / tq2_0 are accepted for CLI compatibility and refused")
    p.add_argument("--verbose", action="store_true")
    p.add_argument("--no_vocab", action="store_true", help="beyond the reference: write tensors and model metadata only")
    return p.parse_args(argv)


def main(argv=None):
    a = parse_args(argv)
    out = convert(a.model, a.dir_model_quant, a.outfile, a.outtype, a.verbose, vocab=not a.no_vocab)
    print(f"Model successfully exported to {out}")


if __name__ == "__main__":
    main()
