"""A tiny sparse-MoE decoder with the module layout of HF's Mixtral up to transformers 4.56 -- the layout the
reference targets (pinned 4.56.0.dev0, README.md:568): `model.layers.<i>.block_sparse_moe.experts.<e>.w1/w2/w3` are
nn.Linear modules, `block_sparse_moe.gate` routes top-k.  (transformers 5.x fuses the experts into 3-D parameters;
there is nothing for a Linear-level quantizer to hook there.)  Test scaffolding only.

Routing is made predictable: token ids < vocab/2 ("A") carry a large component along direction a, the others ("B")
along b; the gate prefers experts {0, 1} for A and {2, 3} for B, and never picks the last expert."""
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

MOE_REGEX = r".*layers.*((q|k|v|o)_proj|experts\.\d+\.w[123])$"


class Attn(nn.Module):
    def __init__(self, h, heads):
        super().__init__()
        self.heads = heads
        self.q_proj, self.k_proj = nn.Linear(h, h, bias=False), nn.Linear(h, h // 2, bias=False)
        self.v_proj, self.o_proj = nn.Linear(h, h // 2, bias=False), nn.Linear(h, h, bias=False)

    def forward(self, x):
        B, L, h = x.shape
        hd = h // self.heads
        q = self.q_proj(x).view(B, L, self.heads, hd).transpose(1, 2)
        k = self.k_proj(x).view(B, L, self.heads // 2, hd).transpose(1, 2).repeat_interleave(2, dim=1)
        v = self.v_proj(x).view(B, L, self.heads // 2, hd).transpose(1, 2).repeat_interleave(2, dim=1)
        o = F.scaled_dot_product_attention(q, k, v, is_causal=True)
        return self.o_proj(o.transpose(1, 2).reshape(B, L, h).contiguous())


class Expert(nn.Module):
    def __init__(self, h, ffn):
        super().__init__()
        self.w1, self.w2, self.w3 = nn.Linear(h, ffn, bias=False), nn.Linear(ffn, h, bias=False), nn.Linear(h, ffn, bias=False)

    def forward(self, x):
        return self.w2(F.silu(self.w1(x)) * self.w3(x))


class SparseMoe(nn.Module):
    def __init__(self, h, ffn, n_experts, top_k):
        super().__init__()
        self.top_k = top_k
        self.gate = nn.Linear(h, n_experts, bias=False)
        self.experts = nn.ModuleList([Expert(h, ffn) for _ in range(n_experts)])

    def forward(self, x):
        B, L, h = x.shape
        flat = x.reshape(-1, h)
        w, sel = torch.topk(F.softmax(self.gate(flat).float(), dim=-1), self.top_k, dim=-1)
        w = (w / w.sum(dim=-1, keepdim=True)).to(x.dtype)
        out = torch.zeros_like(flat)
        for e, expert in enumerate(self.experts):
            tok, slot = torch.where(sel == e)
            if tok.numel() == 0:
                continue  # an expert without tokens is not called: its hooks never fire
            out.index_add_(0, tok, expert(flat[tok]) * w[tok, slot].unsqueeze(-1))
        return out.view(B, L, h)


class Block(nn.Module):
    def __init__(self, h, ffn, heads, n_experts, top_k):
        super().__init__()
        self.input_layernorm, self.post_attention_layernorm = nn.LayerNorm(h), nn.LayerNorm(h)
        self.self_attn = Attn(h, heads)
        self.block_sparse_moe = SparseMoe(h, ffn, n_experts, top_k)

    def forward(self, hidden_states):
        hidden_states = hidden_states + self.self_attn(self.input_layernorm(hidden_states))
        return (hidden_states + self.block_sparse_moe(self.post_attention_layernorm(hidden_states)),)


class _Inner(nn.Module):
    def __init__(self, vocab, h, ffn, heads, n_layers, n_experts, top_k):
        super().__init__()
        self.embed_tokens = nn.Embedding(vocab, h)
        self.layers = nn.ModuleList([Block(h, ffn, heads, n_experts, top_k) for _ in range(n_layers)])
        self.norm = nn.LayerNorm(h)


class TinyMoE(nn.Module):
    def __init__(self, vocab=512, h=256, ffn=512, heads=4, n_layers=2, n_experts=5, top_k=2, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.config = types.SimpleNamespace(use_cache=False, num_local_experts=n_experts, num_experts_per_tok=top_k)
        self.model = _Inner(vocab, h, ffn, heads, n_layers, n_experts, top_k)
        self.lm_head = nn.Linear(h, vocab, bias=False)
        with torch.no_grad():
            for p in self.parameters():
                if p.dim() == 2:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.02)
            a, b = torch.zeros(h), torch.zeros(h)
            a[0], b[1] = 6.0, 6.0
            self.model.embed_tokens.weight[: vocab // 2] += a
            self.model.embed_tokens.weight[vocab // 2:] += b
            for blk in self.model.layers:
                gw = blk.block_sparse_moe.gate.weight
                gw.zero_()
                gw[0, 0], gw[1, 0], gw[2, 1], gw[3, 1] = 4.0, 3.5, 4.0, 3.5   # A -> {0, 1}, B -> {2, 3}
                gw[:4, 2:] = torch.randn(4, h - 2, generator=g) * 0.01
                gw[4:, 0] = -4.0
                gw[4:, 1] = -4.0                                               # the last expert(s): never routed to
        self.eval()

    def forward(self, input_ids):
        x = self.model.embed_tokens(input_ids)
        for blk in self.model.layers:
            x = blk(x)[0]
        return self.lm_head(self.model.norm(x))


def moe_calib(kind, n=4, L=64, vocab=512, seed=3):
    """`n` sequences of [1, L] ids: kind "A" (ids < vocab/2), "B" (ids >= vocab/2) or "AB" (A sequences then B)."""
    g = torch.Generator().manual_seed(seed)
    a = [torch.randint(0, vocab // 2, (1, L), generator=g) for _ in range(n)]
    b = [torch.randint(vocab // 2, vocab, (1, L), generator=g) for _ in range(n)]
    return {"A": a, "B": b, "AB": a + b}[kind]
