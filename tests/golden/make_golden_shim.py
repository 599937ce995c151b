"""Pieces of make_golden.py that the tests need WITHOUT importing the reference
(tiny seeded Llama + calibration ids + the README mixed bit-width map)."""
import torch

MIXED = {"q_proj": "Q3_K", "k_proj": "Q2_K", "v_proj": "Q4_K", "o_proj": "Q5_K", "gate_proj": "Q6_K",
         "down_proj": "Q3_K", "up_proj": "Q4_K", "embed_tokens": "Q6_K", "lm_head": "Q6_K"}


def tiny_llama(seed=0, dtype=torch.float32):
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, vocab_size=512, max_position_embeddings=128, rms_norm_eps=1e-5,
                      tie_word_embeddings=False, attn_implementation="eager")
    torch.manual_seed(seed)
    model = LlamaForCausalLM(cfg).to(dtype)
    model.eval()
    return model


def tiny_calib(n=8, L=64, vocab=512, seed=1):
    g = torch.Generator().manual_seed(seed)
    return [torch.randint(0, vocab, (1, L), generator=g) for _ in range(n)]


OBQ_CASES = (("g128", 48, 512, 128, False, 128), ("g64sym", 32, 256, 64, True, 128), ("g128b64", 32, 256, 128, False, 64))


def obq_inputs(R, C):
    """G14's seeded Linear weight (an all-zero column 7) and calibration inputs (a dead channel 3)."""
    torch.manual_seed(140)
    W = (torch.randn(R, C) * 0.02).half().float()
    W[:, 7] = 0.0
    g = torch.Generator().manual_seed(141)
    sig = torch.exp(torch.randn(C, generator=g) * 0.5)
    xs = [(torch.randn(1, 96, C, generator=g) * sig) for _ in range(3)]
    for x in xs:
        x[..., 3] = 0.0
    return W, xs
