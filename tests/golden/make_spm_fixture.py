#!/usr/bin/env python3
"""Builds tests/golden/spm/tokenizer.model: a tiny SentencePiece model (BPE with byte fallback, like Llama-2 /
TinyLlama / Mixtral: <unk>=0, <s>=1, </s>=2, 256 byte pieces, then learned pieces) trained on a fixed text with the
sentencepiece package of this image.  Data for the packer's vocabulary tests; no reference code involved."""
import io
import os

import sentencepiece as spm

HERE = os.path.dirname(os.path.abspath(__file__))
TEXT = ("the quick brown fox jumps over the lazy dog . quantized integers and bit packed tensors match . "
        "calibration samples shard across the gpus of one node . hessian cholesky inverse trailing update . ") * 40

if __name__ == "__main__":
    model = io.BytesIO()
    spm.SentencePieceTrainer.train(sentence_iterator=iter(TEXT.split(" . ")), model_writer=model, vocab_size=320,
                                   model_type="bpe", byte_fallback=True, character_coverage=1.0, unk_id=0, bos_id=1,
                                   eos_id=2, pad_id=-1, normalization_rule_name="identity", add_dummy_prefix=True,
                                   num_threads=1, seed_sentencepiece_size=1000, user_defined_symbols=["<custom>"])
    os.makedirs(os.path.join(HERE, "spm"), exist_ok=True)
    with open(os.path.join(HERE, "spm", "tokenizer.model"), "wb") as f:
        f.write(model.getvalue())
    print("wrote", len(model.getvalue()), "bytes")
