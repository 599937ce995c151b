"""-m gpu: the opt-in HIP kernels of the calibration forward (csrc/gq_forward.hip, forward_fused.py) against the HF
eager modules they replace (transformers models/llama/modeling_llama.py: LlamaRMSNorm.forward, apply_rotary_pos_emb,
LlamaMLP.forward -- the modules reference quantizer.py:293 runs).

Tolerances, stated here:
  * rotary embedding and silu(gate) * up: every torch op of the eager expression is reproduced with its rounding ->
    bit-exact (torch.equal);
  * RMSNorm: the only freedom is the summation order of mean(x^2) in fp32 (relative 1e-7 on the variance), which can
    move dtype(x * r) across a rounding boundary (1 ulp), which weight * that can stretch to 2 ulp of the 16-bit dtype:
    at most 2 ulp, on fewer than 1 in 1000 elements;
  * a whole decoder layer / a whole Quantizer run: compared as floating point (documented per test).
"""
import os
import sys

import pytest

from conftest import ROOT

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

DTYPES = [torch.bfloat16, torch.float16]


def _ulps(a, b):
    """Distance in representable 16-bit values (same-sign finite inputs assumed where it matters)."""
    ia, ib = a.view(torch.int16).int(), b.view(torch.int16).int()
    ia = torch.where(ia < 0, -(ia & 0x7fff), ia)
    ib = torch.where(ib < 0, -(ib & 0x7fff), ib)
    return (ia - ib).abs()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(1, 2048, 4096), (3, 77, 256), (2, 5, 14336), (4, 64, 8, 128)])
def test_rmsnorm_matches_hf(dtype, shape):
    from transformers.models.llama.modeling_llama import LlamaRMSNorm
    from gptq_gguf_toolkit_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    x = (torch.randn(shape, device="cuda", generator=g) * torch.exp(torch.randn(shape[-1], device="cuda", generator=g))).to(dtype)
    mod = LlamaRMSNorm(shape[-1], eps=1e-5).cuda().to(dtype)
    with torch.no_grad():
        mod.weight.copy_((1.0 + 0.2 * torch.randn(shape[-1], device="cuda", generator=g)).to(dtype))
        want = mod(x)
    got = ops.fwd_rmsnorm(x, mod.weight.data, 1e-5)
    assert got.shape == want.shape and got.dtype == dtype
    d = _ulps(got, want)
    assert int(d.max()) <= 2, f"max {int(d.max())} ulp"
    assert float((d != 0).float().mean()) < 1e-3
    # an all-zero row: rsqrt(eps) * 0
    z = torch.zeros(2, shape[-1], device="cuda", dtype=dtype)
    assert torch.equal(ops.fwd_rmsnorm(z, mod.weight.data, 1e-5), mod(z))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,L,H,Hkv,D", [(1, 2048, 32, 8, 128), (4, 33, 4, 2, 64), (2, 7, 3, 1, 16)])
def test_rope_is_bit_exact(dtype, B, L, H, Hkv, D):
    from transformers.models.llama.modeling_llama import apply_rotary_pos_emb
    from gptq_gguf_toolkit_amd import ops
    g = torch.Generator(device="cuda").manual_seed(6)
    q = torch.randn(B, L, H, D, device="cuda", generator=g).to(dtype)
    k = torch.randn(B, L, Hkv, D, device="cuda", generator=g).to(dtype)
    inv = 1.0 / (500000.0 ** (torch.arange(0, D, 2, device="cuda").float() / D))
    ang = torch.arange(L, device="cuda").float()[None, :, None] * inv[None, None, :]
    emb = torch.cat([ang, ang], -1).expand(B, L, D)
    cos, sin = emb.cos().to(dtype).contiguous(), emb.sin().to(dtype).contiguous()
    wq, wk = apply_rotary_pos_emb(q.transpose(1, 2), k.transpose(1, 2), cos, sin)
    assert torch.equal(ops.fwd_rope(q, cos, sin).transpose(1, 2), wq)
    assert torch.equal(ops.fwd_rope(k, cos, sin).transpose(1, 2), wk)


@pytest.mark.parametrize("dtype", DTYPES)
def test_silu_mul_is_bit_exact(dtype):
    from gptq_gguf_toolkit_amd import ops
    g = torch.Generator(device="cuda").manual_seed(7)
    for shape, scale in (((1, 2048, 14336), 3.0), ((5, 24), 30.0), ((8,), 100.0)):
        gate = (torch.randn(shape, device="cuda", generator=g) * scale).to(dtype)
        up = torch.randn(shape, device="cuda", generator=g).to(dtype)
        assert torch.equal(ops.fwd_silu_mul(gate, up), torch.nn.functional.silu(gate) * up)
    # every finite 16-bit value as the gate
    bits = torch.arange(-32768, 32768, device="cuda", dtype=torch.int32).to(torch.int16)
    gate = bits.view(dtype)
    gate = gate[torch.isfinite(gate.float())]
    gate = gate[: gate.numel() // 8 * 8].contiguous()
    up = torch.full_like(gate, 1.5)
    assert torch.equal(ops.fwd_silu_mul(gate, up), torch.nn.functional.silu(gate) * up)


def test_bad_arguments_fail_loudly():
    from gptq_gguf_toolkit_amd import _cabi, ops
    x = torch.zeros(4, 12, device="cuda", dtype=torch.bfloat16)  # C % 8 != 0
    with pytest.raises(_cabi.GQError):
        ops.fwd_rmsnorm(x, torch.ones(12, device="cuda", dtype=torch.bfloat16), 1e-5)
    with pytest.raises(_cabi.GQError):
        ops.fwd_silu_mul(torch.zeros(4, device="cuda", dtype=torch.bfloat16), torch.zeros(4, device="cuda", dtype=torch.bfloat16))
    with pytest.raises(_cabi.GQError):
        ops.fwd_rmsnorm(torch.zeros(4, 16), torch.ones(16), 1e-5)  # CPU tensors: no fallback
    with pytest.raises(KeyError):
        ops.fwd_silu_mul(torch.zeros(8, device="cuda", dtype=torch.float64), torch.zeros(8, device="cuda", dtype=torch.float64))


@pytest.mark.parametrize("attn", ["eager", "sdpa"])
def test_decoder_layer_under_the_patch(attn):
    """A Llama decoder layer forward with the three modules patched: the Linear hooks see the same number of calls,
    their inputs agree with eager's to bf16 rounding, and the patch is gone after the block."""
    from transformers import LlamaConfig, LlamaForCausalLM
    from transformers.models.llama import modeling_llama as M
    from gptq_gguf_toolkit_amd.forward_fused import fused_forward
    cfg = LlamaConfig(hidden_size=512, intermediate_size=1408, num_hidden_layers=2, num_attention_heads=8,
                      num_key_value_heads=2, vocab_size=512, max_position_embeddings=256, rms_norm_eps=1e-5,
                      tie_word_embeddings=False, attn_implementation=attn)
    torch.manual_seed(0)
    model = LlamaForCausalLM(cfg).to(torch.bfloat16).cuda().eval()
    ids = torch.randint(0, 512, (3, 128), device="cuda")
    seen = {}

    def hook(name, store):
        def fn(mod, args):
            store.setdefault(name, []).append(args[0].detach().float().clone())
        return fn

    def run(store):
        hs = [m.register_forward_pre_hook(hook(n, store)) for n, m in model.named_modules() if isinstance(m, torch.nn.Linear)]
        with torch.no_grad():
            out = model(input_ids=ids, use_cache=False).logits.float()
        for h in hs:
            h.remove()
        return out

    orig = (M.LlamaRMSNorm.forward, M.apply_rotary_pos_emb, M.LlamaMLP.forward)
    eager_in, fused_in = {}, {}
    want = run(eager_in)
    with fused_forward() as patched:
        assert any("LlamaRMSNorm" in p for p in patched) and any("apply_rotary_pos_emb" in p for p in patched)
        assert M.LlamaRMSNorm.forward is not orig[0]
        got = run(fused_in)
    assert (M.LlamaRMSNorm.forward, M.apply_rotary_pos_emb, M.LlamaMLP.forward) == orig
    assert eager_in.keys() == fused_in.keys() and all(len(eager_in[k]) == len(fused_in[k]) == 1 for k in eager_in)
    for k in eager_in:
        a, b = eager_in[k][0], fused_in[k][0]
        # bf16 activations two layers deep: a flipped rounding upstream moves a value by ~2^-8 relative
        assert float((a - b).abs().max()) <= 0.03 * float(a.abs().max()), k
        assert float(((a - b) ** 2).mean().sqrt()) <= 2e-3 * float((a ** 2).mean().sqrt()), k
    assert float((want - got).abs().max()) <= 0.03 * float(want.abs().max())
    with fused_forward(False) as patched:
        assert patched == [] and M.LlamaRMSNorm.forward is orig[0]


def test_quantizer_with_fused_forward(tmp_path):
    """Quantizer(fused_forward=True) on a bf16 model: same tree of files; first-block tensors (whose inputs pass one
    RMSNorm only) nearly all identical, dequantised weights of every Linear close to the eager run's."""
    from make_golden_shim import tiny_calib, tiny_llama
    from gptq_gguf_toolkit_amd.quant_utils import GGMLQuantizationType as T
    from gptq_gguf_toolkit_amd.quantizer import Quantizer
    trees = {}
    orig = {n: p.detach().float() for n, p in tiny_llama(dtype=torch.bfloat16).named_parameters() if p.dim() == 2}
    for fused in (False, True):
        save_dir = str(tmp_path / ("fused" if fused else "eager"))
        os.makedirs(save_dir)
        model = tiny_llama(dtype=torch.bfloat16).cuda()
        data = [([], {"input_ids": ids}) for ids in tiny_calib()]
        drv = Quantizer(model, data_loader=data, quantizable_modules=r".*layers.*((q|k|v|o|gate|up|down)_proj)$",
                        quantizer_kwargs=dict(rel_damp=0.01, block_size=128, act_order=False, quant_scale="absmax",
                                              static_groups=False, rmin=-1.0, rdelta=0.1, nstep=20, verbose=False),
                        pre_block_modules=["model.embed_tokens"], block_modules="model.layers",
                        post_block_modules=["lm_head"], quant_non_block_modules=False, device="cuda:0", save_dir=save_dir,
                        fused_forward=fused)
        drv.quantize({k: T.Q4_K for k in ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")})
        torch.cuda.synchronize()
        assert bool(drv._fused_modules) == fused and drv.fused_forward == ("all" if fused else "off")
        trees[fused] = {n: p.detach().float().cpu() for n, p in model.named_parameters() if p.dim() == 2}
        files = sorted(os.path.relpath(os.path.join(d, f), save_dir) for d, _, fs in os.walk(save_dir) for f in fs)
        trees[(fused, "files")] = files
    assert trees[(False, "files")] == trees[(True, "files")] and len(trees[(True, "files")]) >= 14
    checked = 0
    for n, a in trees[False].items():
        if "layers" not in n:
            continue
        b, w = trees[True][n], orig[n]
        err_e, err_f = float((a - w).norm() / w.norm()), float((b - w).norm() / w.norm())
        # two 4-bit quantisations of one weight from nearly equal Hessians: equally good, and closer to each other
        # than either is to the weight
        assert 0.0 < err_e < 0.2 and abs(err_f - err_e) < 0.1 * err_e, (n, err_e, err_f)
        assert float((a - b).norm() / w.norm()) < err_e, n
        if "layers.0.self_attn.q_proj" in n:
            assert float((a != b).float().mean()) < 0.02, n
        checked += 1
    assert checked == 14


@pytest.mark.parametrize("family", ["mistral", "qwen2", "qwen3"])
def test_other_families_under_the_patch(family):
    """Families whose module text equals Llama's are patched too (forward_fused._targets): logits of a tiny random model
    agree with HF eager to bf16 rounding, and the patch list names the family's classes."""
    import transformers
    from gptq_gguf_toolkit_amd.forward_fused import fused_forward
    cfg_cls, model_cls = {"mistral": ("MistralConfig", "MistralForCausalLM"), "qwen2": ("Qwen2Config", "Qwen2ForCausalLM"),
                          "qwen3": ("Qwen3Config", "Qwen3ForCausalLM")}[family]
    kw = dict(hidden_size=256, intermediate_size=704, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
              vocab_size=512, max_position_embeddings=256, rms_norm_eps=1e-5, tie_word_embeddings=False,
              attn_implementation="sdpa")
    if family == "qwen3":
        kw["head_dim"] = 64
    cfg = getattr(transformers, cfg_cls)(**kw)
    torch.manual_seed(1)
    model = getattr(transformers, model_cls)(cfg).to(torch.bfloat16).cuda().eval()
    ids = torch.randint(0, 512, (2, 96), device="cuda")
    with torch.no_grad():
        want = model(input_ids=ids, use_cache=False).logits.float()
        with fused_forward() as patched:
            assert any(family in p.lower() for p in patched), patched
            got = model(input_ids=ids, use_cache=False).logits.float()
    assert float((want - got).abs().max()) <= 0.03 * float(want.abs().max())
    assert float(((want - got) ** 2).mean().sqrt()) <= 3e-3 * float((want ** 2).mean().sqrt())


def _llama512(dtype):
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = LlamaConfig(hidden_size=512, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, vocab_size=512, max_position_embeddings=128, rms_norm_eps=1e-5,
                      tie_word_embeddings=False, attn_implementation="sdpa")
    torch.manual_seed(0)
    return LlamaForCausalLM(cfg).to(dtype).eval()


@pytest.mark.parametrize("dtype", DTYPES)
def test_default_forward_level_saves_the_same_bytes_as_hf_eager(tmp_path, dtype):
    """The Quantizer's default (fused_forward="exact": rotary embedding, SwiGLU and the order-matched RMSNorm kernel; forward #1
    stopped at the last hooked Linear) against fused_forward=False and against
    the reference's cadence (two full forwards per block, quantizer.py:150-172) on a 16-bit model: every saved tensor and every
    written-back weight is bit-identical -- the defaults change no result.  All three kernels must have been verified and
    used (hidden size 512)."""
    import hashlib
    from make_golden_shim import tiny_calib
    from gptq_gguf_toolkit_amd import forward_fused
    from gptq_gguf_toolkit_amd.quant_utils import GGMLQuantizationType as T
    from gptq_gguf_toolkit_amd.quantizer import Quantizer
    forward_fused._norm_verdict.clear()
    digests = {}
    for level in ("reference_cadence", "off", "exact"):
        save_dir = str(tmp_path / level)
        os.makedirs(save_dir)
        model = _llama512(dtype).cuda()
        data = [([], {"input_ids": ids}) for ids in tiny_calib()]
        kw = {} if level == "exact" else {"fused_forward": False}  # "exact" through the constructor's default
        if level == "reference_cadence":  # r04: the reference's two full forwards per block, nothing overlapped
            kw.update(interrupt_forward1=False)
        drv = Quantizer(model, data_loader=data, quantizable_modules=r".*layers.*((q|k|v|o|gate|up|down)_proj)$",
                        quantizer_kwargs=dict(rel_damp=0.01, block_size=128, act_order=False, quant_scale="absmax",
                                              static_groups=False, rmin=-1.0, rdelta=0.1, nstep=20, verbose=False),
                        pre_block_modules=["model.embed_tokens"], block_modules="model.layers",
                        post_block_modules=["lm_head"], quant_non_block_modules=True, device="cuda:0", save_dir=save_dir, **kw)
        drv.quantize({k: T.Q4_K for k in ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj",
                                          "embed_tokens", "lm_head")})
        torch.cuda.synchronize()
        assert drv.fused_forward == ("exact" if level == "exact" else "off")
        names = " ".join(drv._fused_modules)
        assert (level != "exact" and not names) or all(k in names for k in ("apply_rotary_pos_emb", "LlamaMLP", "LlamaRMSNorm"))
        if level == "reference_cadence":
            assert "forward1_interrupts" not in drv.schedule_stats
        else:  # forward #1 stopped at down_proj for every sample but the first
            assert drv.schedule_stats["forward1_interrupts"] == len(data) - 1
        h = hashlib.sha256()
        for n, p in sorted(model.named_parameters()):
            h.update(p.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes())
        for d, _, fs in sorted(os.walk(save_dir)):
            for f in sorted(fs):
                obj = torch.load(os.path.join(d, f))
                for k in sorted(obj):
                    v = obj[k]
                    h.update(v.contiguous().view(torch.uint8).numpy().tobytes() if torch.is_tensor(v) else str(v).encode())
        digests[level] = h.hexdigest()
    # the three kernels matched HF eager on the run's own first inputs and were used
    assert any(k[:2] == (512, dtype) and v for k, v in forward_fused._norm_verdict.items())
    assert forward_fused._rope_verdict and all(forward_fused._rope_verdict.values())
    assert forward_fused._mlp_verdict and all(forward_fused._mlp_verdict.values())
    assert digests["reference_cadence"] == digests["off"] == digests["exact"]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(1, 2048, 4096), (4, 2048, 4096), (1, 2048, 2048), (1, 2048, 5120), (2, 1024, 8192),
                                   (3, 77, 512), (2, 300, 14336), (1, 64, 3584), (1, 8, 7168)])
def test_rmsnorm_ordered_is_bit_exact(dtype, shape):
    """gq_fwd_rmsnorm_ordered against HF's LlamaRMSNorm: torch.equal (ATen's summation order for the mean, torch.rsqrt's
    correct rounding).  If a PyTorch upgrade changes the reduction order this test fails -- and forward_fused's run-time
    check keeps the eager module, so results stay right either way."""
    from transformers.models.llama.modeling_llama import LlamaRMSNorm
    from gptq_gguf_toolkit_amd import ops
    g = torch.Generator(device="cuda").manual_seed(9)
    C = shape[-1]
    x = (torch.randn(shape, device="cuda", generator=g) * torch.exp(torch.randn(C, device="cuda", generator=g))).to(dtype)
    mod = LlamaRMSNorm(C, eps=1e-5).cuda().to(dtype)
    with torch.no_grad():
        mod.weight.copy_((1.0 + 0.2 * torch.randn(C, device="cuda", generator=g)).to(dtype))
        want = mod(x)
    got, stats = ops.fwd_rmsnorm_ordered(x, mod.weight.data, 1e-5, want_stats=True)
    assert torch.equal(got, want)
    var = x.float().pow(2).mean(-1).reshape(-1)  # the fp32 statistics, bit for bit (the 16-bit outputs hide most differences)
    assert torch.equal(stats[:, 0], var) and torch.equal(stats[:, 1], torch.rsqrt(var + 1e-5))
    z = torch.zeros(8, C, device="cuda", dtype=dtype)  # all-zero rows: rsqrt(eps)
    assert torch.equal(ops.fwd_rmsnorm_ordered(z, mod.weight.data, 1e-5), mod(z))
