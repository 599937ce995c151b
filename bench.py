#!/usr/bin/env python3
"""bench.py -- Mparams/s GPTQ-quantized (BASELINE.json metric) on synthetic Llama-shaped work.

Nothing in this file schedules kernels: the timed step drives the PACKAGE's block scheduler
(`gptq_gguf_toolkit_amd.block_schedule.BlockSchedule`, the object `Quantizer._quant_group` runs) exactly the way
the forward hooks of the drop-in driver do.

Default workload `llama3-8b-block-q4k` (configs[1]): one "step" = one pass of the hot path over ONE Llama-3-8B
transformer block, 218.1 M parameters in 7 Linears, 128 x 2048-token calibration.  Inside the timed region:
  * the hook side: `schedule.feed(name, x)` once per Linear per calibration sequence (the body of the reference's
    forward hook, quantizer.py:226-232 -> GPTQ.update, gptq.py:79-114) and `schedule.sample_done()` per sequence;
    the scheduler buffers the activations and folds them into the 4 distinct Hessians with grouped SYRK launches;
  * `schedule.quantize()`: [N>1] one RCCL all-reduce per distinct Hessian (gptq.py:131-132), then per input group a
    chain on its own HIP stream: fp32 working copy, gq_h_prepare (damping + Cholesky chain), gq_gptq_quantize
    (scale search + column loop + trailing update), gq_dequantize + write-back (quantizer.py:257-264), and -- via
    the scheduler's `extra` callback -- gq_pack (the GGUF block bytes of pack_gptq_into_gguf.py:327-336);
    [N>1] broadcast / all-gather of the results.
Inputs (weights, activations) are resident in HBM before the timed region starts; there is no model forward in
this region (synthetic Linears).

The same JSON line carries, measured AFTER the timed region:
  * `whole_model`: the drop-in pipeline end to end -- a random-init Llama-3-8B-shaped LlamaForCausalLM built on the
    GPU, 128 x 2048 synthetic ids, `Quantizer.quantize` (the region the reference times, quant.py:251-254: capture
    forward, forward #1 + H, solve + column loop, forward #2, RTN of embed/lm_head, data.pth saving) with its split;
    (the Quantizer's default forward: HF modules with the bit-exact HIP kernels -- rotary embedding, SwiGLU, RMSNorm in
    ATen's summation order -- same saved bytes as plain HF eager); `whole_model_hf_eager` the same on plain HF eager
    (--fused_forward off: the reference's forward); `whole_model_batch4` the default with --calibration_batch 4;
  * `trailing_update`: the north star's GEMM three ways (far launches alone, near + far alone, far launches inside
    the timed region);
  * `tolerance_parity` (k_proj, all rows) / `tolerance_parity_widest` (down_proj C = 14336, 128 rows): GPU H -> U -> ints
    against fp64 H -> fp64 chain (on the GPU) -> the oracle's column loop on the same inputs;
  * `cpu_baseline`: the oracle's GPTQ.step on the host cores.

Other workloads (`--workload`): tinyllama-block-q4k, llama3-8b-block-mixed (configs[2]), llama3-70b-block-q4k
(configs[3] shapes), mixtral-block (configs[4] shapes), llama3-8b-model-q4k (a step = the whole model).

`python bench.py --gpus N` with N > 1 and no launcher re-executes itself under torch.distributed.run (one rank per GPU,
RCCL); under a launcher it reads RANK / LOCAL_RANK / WORLD_SIZE.  A rank count other than --gpus is fatal.

Rank 0 prints the full result dict on one line and then, LAST, the compact headline line (<= 2 KB: the driver contract's
keys with `roofline` and `cpu_baseline` flattened to scalars).
"""
import argparse
from pathlib import Path
import json
import os
import shutil
import sys
import tempfile
import time

os.environ.setdefault("OMP_NUM_THREADS", "8")  # cpu_baseline threads (the reference's run_quant.sh:9 pins 8 too)
# One hardware queue per HIP stream: with the runtime's default of 4 two of the block's independent chains share a
# queue and run one after the other (kernel trace: same Queue_Id).  Must be set before the HIP runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from gptq_gguf_toolkit_amd import _cabi, dist_utils, ops  # noqa: E402
from gptq_gguf_toolkit_amd.block_schedule import BlockSchedule  # noqa: E402
from gptq_gguf_toolkit_amd.gptq import GPTQ  # noqa: E402
from gptq_gguf_toolkit_amd.quant_utils import GGMLQuantizationType as QT  # noqa: E402

PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X dense fp16/bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X fp32 matrix (v_mfma_f32_32x32x2_f32: 256 flop/clk/CU x 256 CUs x 2.4 GHz)
QUANTIZER_KW = dict(rel_damp=0.01, block_size=128, act_order=False, quant_scale="absmax", static_groups=False,
                    rmin=-1.0, rdelta=0.1, nstep=20)


def _dense_block(hidden, inter, kv, prefix=""):
    """name -> (R, C, input group): the 7 Linears of a Llama block in module order."""
    return {"q_proj": (hidden, hidden, "attn_in"), "k_proj": (kv, hidden, "attn_in"), "v_proj": (kv, hidden, "attn_in"),
            "o_proj": (hidden, hidden, "o_in"), "gate_proj": (inter, hidden, "mlp_in"),
            "up_proj": (inter, hidden, "mlp_in"), "down_proj": (hidden, inter, "down_in")}


def _mixtral_block():
    s = {"q_proj": (4096, 4096, "attn_in"), "k_proj": (1024, 4096, "attn_in"), "v_proj": (1024, 4096, "attn_in"),
         "o_proj": (4096, 4096, "o_in")}
    for e in range(8):  # HF <= 4.56 module names: block_sparse_moe.experts.<e>.w1/w3 (in) and w2 (out)
        s[f"experts.{e}.w1"] = (14336, 4096, f"e{e}_in")
        s[f"experts.{e}.w3"] = (14336, 4096, f"e{e}_in")
        s[f"experts.{e}.w2"] = (4096, 14336, f"e{e}_mid")
    return s


MIXED = {"q_proj": "Q3_K", "k_proj": "Q2_K", "v_proj": "Q4_K", "o_proj": "Q5_K", "gate_proj": "Q6_K",
         "down_proj": "Q3_K", "up_proj": "Q4_K"}  # the reference README's mixed map (README.md:94-106)
WORKLOADS = {
    "llama3-8b-block-q4k": dict(shapes=_dense_block(4096, 14336, 1024), nseq=128, L=2048, q="Q4_K"),
    "llama3-8b-block-mixed": dict(shapes=_dense_block(4096, 14336, 1024), nseq=128, L=2048, q=MIXED),
    "tinyllama-block-q4k": dict(shapes=_dense_block(2048, 5632, 256), nseq=32, L=512, q="Q4_K"),
    "llama3-70b-block-q4k": dict(shapes=_dense_block(8192, 28672, 1024), nseq=128, L=4096, q="Q4_K"),
    "mixtral-block": dict(shapes=_mixtral_block(), nseq=128, L=2048,
                          q={"q_proj": "Q6_K", "k_proj": "Q6_K", "v_proj": "Q6_K", "o_proj": "Q6_K", "w1": "Q3_K",
                             "w2": "Q3_K", "w3": "Q3_K"}, experts=8, top_k=2),
    "llama3-8b-model-q4k": dict(model=dict(hidden_size=4096, intermediate_size=14336, num_hidden_layers=32,
                                           num_attention_heads=32, num_key_value_heads=8, vocab_size=128256,
                                           max_position_embeddings=8192, rope_theta=500000.0, rms_norm_eps=1e-5),
                                nseq=128, L=2048, q="Q4_K"),
}


# ----------------------------------------------------------------------------- synthetic inputs
def make_inputs(wl, nseq, L, dev, seed=1):
    """-> {input group: [per-sequence activation tensors]}.  X ~ N(0,1) * sigma_c, sigma_c log-normal, 0.1 % outlier
    channels x20, fp16.  Dense inputs are [1, L, C] per sequence; expert inputs are the [tokens, C] rows routed to
    that expert (top_k of `experts` drawn uniformly per token: about L * top_k / experts rows per sequence)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    X = {}
    experts, top_k = wl.get("experts"), wl.get("top_k")
    counts = None
    if experts:
        scores = torch.rand(nseq, L, experts, device=dev, generator=g)
        picked = torch.zeros_like(scores, dtype=torch.bool).scatter_(2, scores.topk(top_k, dim=2).indices, True)
        counts = picked.sum(dim=1).cpu()  # [nseq, experts]
    for name, (R, C, inp) in wl["shapes"].items():
        if inp in X:
            continue
        sig = torch.exp(torch.randn(C, device=dev, generator=g) * 0.5)
        sig[torch.randperm(C, device=dev, generator=g)[:max(1, C // 1000)]] *= 20.0
        rows = [L] * nseq if not inp.startswith("e") or counts is None else counts[:, int(inp[1:].split("_")[0])].tolist()
        pool = torch.empty(sum(rows), C, device=dev, dtype=torch.float16)
        o = 0
        for n in rows:  # generated in slices: an fp32 [T, C] temporary would not fit next to the rest at 70B sizes
            for a in range(0, n, 8192):
                b = min(n, a + 8192)
                pool[o + a:o + b] = (torch.randn(b - a, C, device=dev, generator=g) * sig).half()
            o += n
        parts, o = [], 0
        for n in rows:
            parts.append(pool[o:o + n].unsqueeze(0) if counts is None or not inp.startswith("e") else pool[o:o + n])
            o += n
        X[inp] = parts
    return X


def make_layers(shapes, dev, seed=0):
    """nn.Linear modules (fp16, W ~ N(0, 0.02^2)) + the pristine weights (a step's write-back replaces weight.data)."""
    layers, W16 = {}, {}
    for i, (name, (R, C, _)) in enumerate(shapes.items()):
        g = torch.Generator(device=dev).manual_seed(seed + i)
        lin = nn.Linear(C, R, bias=False, device="meta")
        W16[name] = (torch.randn(R, C, device=dev, generator=g) * 0.02).half()
        lin.weight = nn.Parameter(W16[name], requires_grad=False)
        layers[name] = lin
    return layers, W16


def q_of(wl, name):
    q = wl["q"]
    return QT[q] if isinstance(q, str) else QT[q[name.split(".")[-1]]]


def block_step(wl, layers, W16, X, keep=None):
    """One step THROUGH THE PACKAGE: hook-side feeding, then BlockSchedule.quantize.  Returns the schedule."""
    for name, lin in layers.items():
        lin.weight.data = W16[name]
    sched = BlockSchedule(layers, lambda l, n: GPTQ(l, allow_no_samples=".experts." in f".{n}.", **QUANTIZER_KW))
    names = list(layers)
    nseq = len(next(iter(X.values())))
    t_host = time.perf_counter()
    for s in range(nseq):
        for name in names:  # module order, like the forward: q, k, v see the very same tensor object
            x = X[wl["shapes"][name][2]][s]
            if x.shape[-2] > 0:
                sched.feed(name, x)
        sched.sample_done()
    qtypes = {n: q_of(wl, n) for n in names}

    def extra(name, h, res):
        packed = ops.pack(int(qtypes[name]), *res)
        if keep is not None:
            keep[name] = (res, packed, h._last_U)
        return packed

    t_fed = time.perf_counter()
    out = sched.quantize(qtypes, writeback=True, extra=extra)
    if os.environ.get("GQ_BENCH_TRACE_HOST"):
        print(f"  [host] feed {1e3 * (t_fed - t_host):.2f} ms, quantize() {1e3 * (time.perf_counter() - t_fed):.2f} ms "
              f"(returns after the block's one host sync)", file=sys.stderr)
    if keep is not None:
        keep["__out__"] = out
    return sched


def syrk_flops(wl, X):
    """Algorithmic MFMA flops of one step's Hessian accumulation: upper-triangular 128x128 tiles,
    T * C * (C + 128) per distinct input."""
    f, seen = 0.0, set()
    for name, (R, C, inp) in wl["shapes"].items():
        if inp not in seen:
            seen.add(inp)
            f += float(sum(x.shape[-2] for x in X[inp])) * C * (C + 128)
    return f


# ----------------------------------------------------------------------------- side legs (after the timed region)
def trailing_update_legs(wl, W16, X, in_region):
    """The blocked trailing-update GEMM (north star: >= 70 % of the fp32 MFMA peak) of the block's widest Linear,
    with the U of a real gq_h_prepare on that Linear's Hessian:
      far_alone     the chained far updates (one per 1024-column super-block), alone on the GPU
      whole_alone   near (rest-of-super-block after every 128-column block) + far launches, alone on the GPU
                    (both with the helper stream off: every GEMM has the chip to itself -- the kernels' own efficiency)
      loop_ms       the Linear's whole column loop, alone: one stream, and as the product runs it (far updates cut
                    by column groups and moved next to the loop on the library's helper stream, persistent launches
                    with 192 workgroups: each far GEMM is slower, the loop as a whole shorter)
      far_in_region the far launches as they ran INSIDE the timed region, next to the other chains
    Algorithmic flops: far = sum over super-blocks 2 R (S1-S0)(C-S1); near = sum over blocks 2 R 128 (S1-c2)."""
    try:
        shapes = wl["shapes"]
        name = max(shapes, key=lambda n: shapes[n][0] * shapes[n][1] * shapes[n][1])
        R, C, inp = shapes[name]
        dev = W16[name].device
        H = torch.zeros(C, C, device=dev)
        xs = torch.cat([x.reshape(-1, C) for x in X[inp][:8]])
        ops.h_accumulate(H, xs, 0.0, 2.0 / 8)
        U, _ = ops.h_prepare(H, W16[name].float(), 0.01)
        del H, xs
        B = 128
        sb = B * int(ops.option_get("la"))  # columns per look-ahead super-block (gq_gptq.hip LA)
        far = sum(2.0 * R * (min(s0 + sb, C) - s0) * (C - min(s0 + sb, C)) for s0 in range(0, C, sb))
        near_all = sum(2.0 * R * B * (min((c1 // sb + 1) * sb, C) - (c1 + B)) for c1 in range(0, C, B))
        if ops.option_get("near_classic"):
            near, fused = near_all, 0.0
        else:
            # r03: an even block's errors reach its partner block inside the column-loop kernel (not a GEMM launch: its
            # flops are NOT counted here); the launches are the chained K = 256 updates after every 256-column group
            # (a column of a later pair receives the same blocks' errors whether they arrive pair by pair or quad by quad:
            # the launched flops are those of the pair-wise form)
            near = sum(2.0 * R * 2 * B * (min((c1 // sb + 1) * sb, C) - (c1 + 2 * B)) for c1 in range(0, C, 2 * B))
            fused = near_all - near
        best, loop_ms = {}, {}
        helper_was = ops.far_helper_enable(True)
        try:
            for mode in ("one_stream", "as_run"):
                ops.far_helper_enable(mode == "as_run")
                for it in range(3):  # un-profiled wall time of the whole loop
                    Wf = W16[name].float()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    ops.gptq_quantize(Wf, U, int(q_of(wl, name)), B)
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t0) * 1e3
                    loop_ms[mode] = min(loop_ms.get(mode, dt), dt)
                if mode != "one_stream":
                    continue
                for it in range(2):
                    Wf = W16[name].float()
                    torch.cuda.synchronize()
                    _cabi.prof_enable(["trailing_far_gemm32", "trailing_gemm32"])
                    ops.gptq_quantize(Wf, U, int(q_of(wl, name)), B)
                    torch.cuda.synchronize()
                    got = _cabi.prof_collect(busy=True)
                    _cabi.prof_enable([])
                    for k, v in got.items():
                        if k not in best or v[0] < best[k][0]:
                            best[k] = v
        finally:
            ops.far_helper_enable(helper_was)
        fms, fn, _ = best.get("trailing_far_gemm32", (0.0, 0, 0.0))
        nms, nn_, _ = best.get("trailing_gemm32", (0.0, 0, 0.0))
        out = {"bound": "mfma", "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "linear": f"{name} {R}x{C}",
               "U": "gq_h_prepare of this Linear's Hessian (8 calibration sequences)",
               "kernels": "gemm32_chain_full_kernel<128> (far, K = 1024); gemm32_near256_kernel (near: K = 256 after every 256-column group)"}
        if fn:
            a = far / (fms * 1e-3) / 1e12
            out["far_alone"] = {"achieved": round(a, 2), "frac": round(a / PEAK_F32_MFMA_TFLOPS, 4), "launches": fn,
                                "ms": round(fms, 3)}
        if fn and nn_:
            a = (far + near) / ((fms + nms) * 1e-3) / 1e12
            out["whole_alone"] = {"achieved": round(a, 2), "frac": round(a / PEAK_F32_MFMA_TFLOPS, 4),
                                  "launches": fn + nn_, "ms": round(fms + nms, 3), "near_ms": round(nms, 3),
                                  "near_GFLOP": round(near / 1e9, 1), "far_GFLOP": round(far / 1e9, 1),
                                  "GFLOP_inside_the_column_loop_kernel_not_counted": round(fused / 1e9, 1)}
        out["loop_ms"] = {"one_stream": round(loop_ms["one_stream"], 2), "as_run": round(loop_ms["as_run"], 2),
                          "far_updates_on_helper_stream": bool(ops.uses_helper_stream(R, C, B))}
        if in_region and in_region[1]:
            # launches of EVERY Linear of the block ran under this tag in the region: price them all
            far_all = sum(sum(2.0 * r * (min(s0 + sb, c) - s0) * (c - min(s0 + sb, c)) for s0 in range(0, c, sb))
                          for r, c, _ in shapes.values())
            a = far_all * in_region[3] / (in_region[0] * 1e-3) / 1e12
            out["far_in_region"] = {"achieved": round(a, 2), "frac": round(a / PEAK_F32_MFMA_TFLOPS, 4),
                                    "launches": in_region[1], "ms_per_step": round(in_region[0] / in_region[3], 3),
                                    "note": "all Linears of the block, sum of launch durations, other chains running"}
            if in_region[2]:
                # the chains overlap: two far GEMMs that share the chip each take twice as long, and the SUM of their durations
                # counts that time twice.  Over the UNION of the launch intervals (how the SYRK's roofline is taken):
                u = far_all * in_region[3] / (in_region[2] * 1e-3) / 1e12
                out["far_in_region"].update({"achieved_over_busy_time": round(u, 2), "frac_over_busy_time": round(u / PEAK_F32_MFMA_TFLOPS, 4),
                                             "busy_ms_per_step": round(in_region[2] / in_region[3], 3)})
        return out
    except Exception as e:  # the bench line must still print
        return {"error": repr(e)}


TYPE_SIZE = {10: 84, 11: 110, 12: 144, 13: 176, 14: 210}  # bytes per 256-value block
PEAK_HBM_TBPS = 8.0


def encoder_legs(wl, layers, W16, X, tu):
    """SURVEY section 8(d)'s remaining figures (VERDICT r03 missing #4), from ONE more step of the same schedule after the
    timed region with HIP events on the codec kernels' launch streams (the events cost ~4 us per launch on a chain of
    ~170 launches, so they are kept out of the K timed steps; the other chains run next to them as in the region):
      encoders     the HBM figure of the encoder side: sum over the block's Linears of R C (2 + 1 + type_size/256 + 2 + 4 C / R)
                   bytes -- W fp16 in, ints out, packed out, dequantized fp16 out, U once -- (5.6 B/param + 4C/R for Q4_K)
                   over the union of the launch intervals of scale_search* / gptq_segment / dequantize / pack (the column
                   loop's own kernels are codec work: K4 + K5 + K7 + K9-13), and dequantize / pack alone with their own bytes
                   (1 + 2 and 1 + type_size/256 per param);
      column_loop  dependent column steps per second of the widest Linear's loop (alone on the GPU, as the product runs
                   it: trailing_update.loop_ms.as_run) and the column-loop kernel's launch time per 128-column block."""
    try:
        tags = ["scale_search", "gptq_segment", "dequantize", "pack"]
        _cabi.prof_enable(tags)
        block_step(wl, layers, W16, X)
        torch.cuda.synchronize()
        got = _cabi.prof_collect(busy=True)
        _cabi.prof_enable([])
        shapes = wl["shapes"]
        byts = sum(float(R) * C * (2 + 1 + TYPE_SIZE[int(q_of(wl, n))] / 256 + 2 + 4.0 * C / R) for n, (R, C, _) in shapes.items())
        params = sum(float(R) * C for R, C, _ in shapes.values())
        tot_ms = sum(got[t][0] for t in tags if t in got)  # sum of launch durations (chains overlap: not a wall time)
        enc = {"bound": "hbm", "peak": PEAK_HBM_TBPS, "unit": "TB/s",
               "bytes_per_param": round(byts / params, 3), "GB_per_step": round(byts / 1e9, 3),
               "kernel_ms_per_step": {t: round(got[t][0], 3) for t in tags if t in got},
               "launches": {t: got[t][1] for t in tags if t in got},
               "achieved": round(byts / (tot_ms * 1e-3) / 1e12, 4) if tot_ms else None,
               "frac": round(byts / (tot_ms * 1e-3) / 1e12 / PEAK_HBM_TBPS, 5) if tot_ms else None,
               "note": "sum of launch durations over the four chains; the column loop is latency-bound by C dependent steps "
                       "(column_loop), not by these bytes"}
        for t, per in (("dequantize", lambda n: 3.0), ("pack", lambda n: 1 + TYPE_SIZE[int(q_of(wl, n))] / 256)):
            if t in got and got[t][0]:
                b = sum(float(R) * C * per(n) for n, (R, C, _) in shapes.items())
                a = b / (got[t][0] * 1e-3) / 1e12
                enc[t] = {"achieved": round(a, 3), "frac": round(a / PEAK_HBM_TBPS, 4), "bytes_per_param": round(b / params, 4)}
        name = max(shapes, key=lambda n: shapes[n][0] * shapes[n][1] * shapes[n][1])
        R, C, _ = shapes[name]
        col = {"linear": f"{name} {R}x{C}", "steps": C}
        loop = (tu or {}).get("loop_ms", {}).get("as_run")
        if loop:
            col.update({"loop_ms_alone": loop, "steps_per_s": round(C / (loop * 1e-3), 0), "ns_per_step": round(loop * 1e6 / C, 1)})
        if "gptq_segment" in got and got["gptq_segment"][1]:
            col["segment_kernel_us_per_launch_in_step"] = round(got["gptq_segment"][0] * 1e3 / got["gptq_segment"][1], 2)
            col["all_linears_steps_per_s_of_segment_kernel_time"] = round(
                sum(C_ for _, C_, _ in shapes.values()) / (got["gptq_segment"][0] * 1e-3), 0)
        return enc, col
    except Exception as e:
        return {"error": repr(e)}, {"error": repr(e)}


def tolerance_parity(wl, W16, X, n_seq=8, widest=False, n_rows=None):
    """K1/K3 are tolerance-class (summation order).  End-to-end effect on the result, on one Linear: the GPU's
    H -> U -> ints against an fp64 H -> fp64 LAPACK Cholesky chain -> the oracle's column loop (BASELINE.md section
    3): share of differing ints and scale bytes, max |delta w_hat|.  `ulp_noise_floor` is the same comparison
    between two ORACLE runs whose U differ by a 1e-7 relative perturbation (less than one fp32 rounding): the
    column loop's error feedback amplifies any last-bit difference, so this is the rate two correct fp32
    implementations differ by.  Bounded: `n_seq` calibration sequences; the k_proj rows, or -- `widest`: the Linear with
    the widest input (down_proj, C = 14336: the only chain on which all three image levels of K3 run) -- `n_rows` rows
    of it through the oracle.  The fp64 Hessian and chain run on the GPU (torch: hipBLAS / hipSOLVER); the checker's
    column loop is the oracle's C restatement on the host."""
    try:
        from oracle import oracle as O
        shapes = wl["shapes"]
        if widest:
            name = max(shapes, key=lambda n: (shapes[n][1], shapes[n][0]))
        else:
            name = "k_proj" if "k_proj" in shapes else min(shapes, key=lambda n: shapes[n][0] * shapes[n][1])
        R, C, inp = shapes[name]
        q_type = int(q_of(wl, name))
        xs = torch.cat([x.reshape(-1, C) for x in X[inp][:n_seq]])
        Wf = W16[name].float()
        H = torch.zeros(C, C, device=Wf.device)
        ops.h_accumulate(H, xs, 0.0, 2.0 / n_seq)
        Wg = Wf.clone()
        U, flag = ops.h_prepare(H.clone(), Wg, 0.01)
        q, d, s, dmin, m = ops.gptq_quantize(Wg, U, q_type, 128)
        # the same with the chain's level-3 work on the fp32 matrix instruction throughout: the REFERENCE's precision
        # (linalg_utils.py:8-12 in fp32) through the same kernels -- the yardstick for "is the default chain good enough"
        with ops.options(chol_fp32=1, chol_3p_min=0):
            W32 = Wf.clone()
            U32c, _ = ops.h_prepare(H.clone(), W32, 0.01)
            q32 = ops.gptq_quantize(W32, U32c, q_type, 128)[0]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        H64 = torch.zeros(C, C, device=Wf.device, dtype=torch.float64)
        for i in range(0, xs.shape[0], 4096):  # bounded fp64 scratch
            x64 = xs[i:i + 4096].double()
            H64.addmm_(x64.T, x64, alpha=2.0 / n_seq)
        del x64
        h_err = float((H.double() - H64).abs().max() / H64.abs().max())
        H64.diagonal().add_(0.01 * H64.diagonal().mean())  # gptq.py:304-324 in fp64 (no dead channel / zero column here)
        L = torch.linalg.cholesky(H64)
        del H64
        Hi = torch.cholesky_inverse(L)
        del L
        Uo_d = torch.linalg.cholesky(Hi, upper=True)
        del Hi
        u_err = float((U.double() - Uo_d).abs().max() / Uo_d.abs().max())
        u_err32 = float((U32c.double() - Uo_d).abs().max() / Uo_d.abs().max())
        del U32c
        Uo = Uo_d.cpu().numpy()
        del Uo_d
        rows = slice(0, R) if not n_rows or n_rows >= R else slice(R // 4, R // 4 + n_rows)
        Wo = Wf[rows].cpu().numpy()
        U32 = Uo.astype(np.float32)
        Wd, oq, od, os_, odm, om = O.gptq_step(Wo, U32, q_type, block_size=128)
        rng = np.random.default_rng(0)
        Un = (Uo * (1.0 + 1e-7 * rng.standard_normal(Uo.shape))).astype(np.float32)
        _, nq, nd, ns, ndm, nm = O.gptq_step(Wo, Un, q_type, block_size=128)
        dt = time.perf_counter() - t0
        bits = lambda t: t[rows].cpu().view(torch.int16).numpy().view(np.uint16)  # noqa: E731

        def scale_rate(a, b):
            return float(np.concatenate([(x != y).ravel() for x, y in zip(a, b)]).mean())

        return {"linear": f"{name} {R}x{C} {QT(q_type).name}", "rows_checked": int(rows.stop - rows.start),
                "tokens": int(xs.shape[0]), "H_rel_err": h_err, "U_rel_err": u_err,
                "ints_differ": float((q[rows].cpu().numpy() != oq).mean()),
                "scale_bytes_differ": scale_rate((bits(d), s[rows].cpu().numpy(), bits(dmin), m[rows].cpu().numpy()),
                                                 (od, os_, odm, om)),
                "max_abs_dw": float(np.abs(Wg[rows].cpu().numpy() - Wd).max()),
                "ulp_noise_floor": {"ints_differ": float((nq != oq).mean()),
                                    "scale_bytes_differ": scale_rate((nd, ns, ndm, nm), (od, os_, odm, om))},
                "all_fp32_chain": {"U_rel_err": u_err32, "ints_differ": float((q32[rows].cpu().numpy() != oq).mean()),
                                   "note": "the same Linear with the chain's GEMMs on v_mfma_f32_32x32x2_f32 (options chol_fp32, "
                                           "chol_3p_min = 0): the reference's precision; the default (16-bit images) is at least as close "
                                           "to the fp64 result"},
                "checker_s": round(dt, 1),
                "vs": "fp64 H and fp64 Cholesky chain (torch on the GPU), the oracle's C restatement of GPTQ.step on the host"}
    except Exception as e:
        return {"error": repr(e)}


def fast_obq_leg(wl, W16, X, n_seq=32, bits=(2, 3, 4, 8)):
    """f4: EvoPress' uniform-grid GPTQ (evopress/src/fast_obq.py) on the block's widest Linear through the package's
    FastOBQ handle -- one Hessian and ONE factorisation, then one column loop per bit width (the layer database the
    search reads, evopress/src/quantizer.py:146-171) -- timed alone on the GPU (parity: tests/test_gpu_obq.py)."""
    try:
        from gptq_gguf_toolkit_amd.fast_obq import FastOBQ
        shapes = wl["shapes"]
        name = max(shapes, key=lambda n: shapes[n][0] * shapes[n][1] * shapes[n][1])
        R, C, inp = shapes[name]
        lin = torch.nn.Linear(C, R, bias=False, device=W16[name].device, dtype=W16[name].dtype)
        lin.weight.data = W16[name]
        out = {}
        for it in range(2):  # the second pass is the timed one
            h = FastOBQ(lin, bitwidth_options=list(bits), group_size=128, sym=False, rel_damp=0.01, block_size=128)
            for x in X[inp][:n_seq]:
                h.update(x)
            h.flush()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            q, sc, ze, _ = h.quantize(list(bits))
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        out = {"linear": f"{name} {R}x{C}", "bit_widths": list(bits), "group_size": 128, "tokens": int(n_seq * X[inp][0].numel() // C),
               "ms_prepare_plus_all_loops": round(dt * 1e3, 2),
               "Mparams_per_s_per_bit_width": round(R * C * len(bits) / dt / 1e6, 1),
               "parity": "tests/test_gpu_obq.py (bit-exact against the reference's own run G14 and the oracle at this shape)",
               "note": "one factorisation, len(bit_widths) column loops; timed alone on the GPU after the timed region"}
        h.reset()
        return out
    except Exception as e:
        return {"error": repr(e)}


def cpu_baseline(wl, W16, keep):
    """The path on the host cores, every stage measured in THIS run on a bounded sample (~15-25 s of CPU):
      * GPTQ.step (gptq.py:145-276) of every Linear of the block through the oracle's C restatement (OpenMP), with the U
        the GPU used -- measured in full, ints compared with the GPU's;
      * dequantize + pack of the same Linears (quant_utils.py:277-310, packing_utils.py:33-326) through the oracle;
      * GPTQ.update (gptq.py:96-112: `H.addmm_(X.T, X)`, fp32) and GPTQ._prepare's chain (:318-320: cholesky ->
        cholesky_inverse -> cholesky(upper)) AS THE REFERENCE RUNS THEM ON A CPU: the same torch calls (MKL underneath) on
        a sample -- 16 384 tokens x 4096 channels in 2048-token updates, one 4096-wide chain -- scaled to the block by
        flops (the reference accumulates and factorises one Hessian per Linear: sum of 2 T C^2 and of C^3 over the 7
        Linears).  (The oracle's own h_accumulate / h_prepare are fp64 accuracy anchors with naive loops, not what a CPU
        user would run.)
    `value` is the full path; the step-only figure of r01/r02 stays beside it."""
    try:
        from oracle import oracle as O
        shapes = wl["shapes"]
        names, budget = [], 2.0e12  # bounded: sum of R C^2 (one Llama-3-8B block: 1.6e12, ~10 s on 8 cores)
        for n in shapes:
            cost = float(shapes[n][0]) * shapes[n][1] ** 2
            if n in keep and keep[n][2] is not None and cost <= budget:
                names.append(n)
                budget -= cost
        threads = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
        tot, same, cnt, dt, dt_codec = 0, 0.0, 0, 0.0, 0.0
        for n in names:
            R, C, _ = shapes[n]
            qt = int(q_of(wl, n))
            W = W16[n].float().cpu().numpy()
            U = keep[n][2].cpu().numpy()
            t0 = time.perf_counter()
            _, oq, od, os_, odm, om = O.gptq_step(W, U, qt, block_size=128)
            t1 = time.perf_counter()
            O.dequantize(qt, oq, od, os_, odm, om)
            O.pack(qt, oq, od, os_, odm, om)
            dt_codec += time.perf_counter() - t1
            dt += t1 - t0
            tot += R * C
            same += float((oq == keep["__out__"][n][0].cpu().numpy()).sum())
            cnt += oq.size
            del W, U, oq
        # the reference's own CPU calls for the two tolerance-class stages, on a sample
        old_threads = torch.get_num_threads()
        torch.set_num_threads(threads)
        try:
            # GPTQ.update measured at EVERY distinct width of the block (a few sequences each: ~1 s per width), scaled by the
            # token count only; the chain at the narrowest width, scaled by C^3
            g = torch.Generator().manual_seed(0)
            Ls = 2048
            widths = sorted({shapes[n][1] for n in names})
            t_upd, upd_note = {}, []
            for Cw in widths:
                nsq = max(1, min(8, int(8 * (4096.0 / Cw) ** 2 + 0.5)))
                Xs = [torch.randn(Ls, Cw, generator=g).half() for _ in range(nsq)]
                H = torch.zeros(Cw, Cw)
                t0 = time.perf_counter()
                for i, x in enumerate(Xs):  # gptq.py:96-112 per calibration sample
                    xf = x.float()
                    H.addmm_(xf.T, xf, beta=i / (i + 1), alpha=2.0 / (i + 1))
                t_upd[Cw] = (time.perf_counter() - t0) / (nsq * Ls)  # seconds per token at this width
                upd_note.append(f"{nsq} x {Ls} tokens x {Cw} channels {t_upd[Cw] * nsq * Ls:.2f} s")
                if Cw == widths[0]:
                    Cs = Cw
                    H.diagonal().add_(0.01 * H.diagonal().mean())
                    t0 = time.perf_counter()
                    torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(H)), upper=True)  # :318-320
                    t_c = time.perf_counter() - t0
                del Xs, H
        finally:
            torch.set_num_threads(old_threads)
        T_full = wl["nseq"] * wl["L"]
        t_h_total = sum(float(T_full) * t_upd[shapes[n][1]] for n in names)  # the reference accumulates one Hessian per Linear
        c_scale = sum(float(shapes[n][1]) ** 3 for n in names) / float(Cs) ** 3
        est = {"update_s": t_h_total, "prepare_chain_s": t_c * c_scale, "step_s": dt, "dequantize_pack_s": dt_codec}
        total = sum(est.values())
        return {"value": round(tot / total / 1e6, 4), "unit": "Mparams/s", "cores": threads, "kind": "port",
                "sample": f"full path of {len(names)} of the block's {len(shapes)} Linears ({'/'.join(names)}, {tot / 1e6:.1f} M "
                          f"params): GPTQ.step {dt:.1f} s and dequantize + pack {dt_codec:.1f} s measured in full (the oracle's C "
                          f"restatement; ints equal to the GPU's: {same / cnt:.6f}); GPTQ.update measured as the reference's "
                          f"own fp32 addmm at every width of the block ({'; '.join(upd_note)}) and scaled by TOKENS only to the "
                          f"block's {len(names)} Hessians of {T_full} tokens; the Cholesky chain measured as "
                          f"the reference's torch calls at C = {Cs} ({t_c:.2f} s) and scaled x{c_scale:.1f} by C^3",
                "stages_s_per_block": {k: round(v, 2) for k, v in est.items()},
                "step_only": {"value": round(tot / dt / 1e6, 3), "unit": "Mparams/s",
                              "note": "GPTQ.step alone (r01/r02's figure): scale search + column loop + trailing update"}}
    except Exception as e:  # the bench line must still print
        return {"value": None, "unit": "Mparams/s", "cores": 0, "kind": "port", "sample": f"failed: {e!r}"}


# ----------------------------------------------------------------------------- whole model (the drop-in pipeline)
def build_model(cfg_kw, dev, dtype=torch.bfloat16, seed=0):
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = LlamaConfig(tie_word_embeddings=False, attn_implementation=os.environ.get("GQ_ATTN", "sdpa"), **cfg_kw)
    torch.manual_seed(seed)
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        with torch.device(dev):
            model = LlamaForCausalLM(cfg)
    finally:
        torch.set_default_dtype(old)
    model.eval()
    return model


def gguf_leg(hf_dir, save_dir, root, quant_wall):
    """The second half of the target (north star: "Llama-3-8B -> Q4_K end-to-end ... GGUF output"): pack_gptq_into_gguf.convert on
    the data.pth tree the quantizer has just written + the checkpoint directory, into one .gguf file (reference
    pack_gptq_into_gguf.py:282-349, 469-475).  vocab=False: a random-init model has no tokenizer files (the vocabulary is a few MB
    of metadata: tests/test_host_logic_cpu.py covers it)."""
    from gptq_gguf_toolkit_amd.pack_gptq_into_gguf import convert
    out = {}
    flows = [("pipelined", True)] + ([("tensor_by_tensor", False)] if os.environ.get("GQ_BENCH_GGUF_FLOW") == "1" else [])
    for label, pipelined in flows:
        outfile = os.path.join(root, f"gq_bench_{os.getpid()}_{label}.gguf")
        tm = {}
        try:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            convert(Path(hf_dir), Path(save_dir), Path(outfile), "f16", vocab=False, pipelined=pipelined, timing=tm)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            size = os.path.getsize(outfile)
            digest = None
            if os.environ.get("GQ_BENCH_GGUF_FLOW") == "1":
                import hashlib
                hsh = hashlib.sha256()
                with open(outfile, "rb") as f:
                    for chunk in iter(lambda: f.read(1 << 26), b""):
                        hsh.update(chunk)
                digest = hsh.hexdigest()[:16]
        finally:
            if os.path.exists(outfile):
                os.remove(outfile)
        out[label] = {"pack_wall_s": round(wall, 2), "gguf_GB": round(size / 1e9, 2), "sha256_16": digest,
                      "split_s": {k: round(v, 2) for k, v in sorted(tm.items())},
                      "split_keys": "load = torch.load of the data.pth files (mmap); hf_read / plain = checkpoint tensors GPTQ did not "
                                    "replace, read and converted; h2d / permute_pack / d2h = producer threads, seconds ADDED UP over the "
                                    "GGUFWriter.LAZY_WORKERS = 3 that run at a time (upload of the five tensors, q/k row un-permute + gq_pack "
                                    "on the GPU incl. waiting for the other producers' work on the stream, download of the block bytes); write = "
                                    "file writes, wait = the writing thread waiting for a producer; write_call = GGUFWriter.write as a whole (wall)"}
    p = out["pipelined"]
    return {"end_to_end_gguf_wall_s": round(quant_wall + p["pack_wall_s"], 2), "quantize_wall_s": round(quant_wall, 2),
            "pack_wall_s": p["pack_wall_s"], "gguf_GB": p["gguf_GB"], "split_s": p["split_s"], "split_keys": p["split_keys"],
            "flows": out if len(out) > 1 else None, "outtype": "f16 for what GPTQ did not quantize; vocab=False",
            "region": "Quantizer.quantize (quant.py:251-254) + pack_gptq_into_gguf.convert (pack_gptq_into_gguf.py:282-349, 469-475) back to "
                      "back; checkpoint directory (safetensors) written before the timed region"}


def whole_model_run(wl, dev, world, rank, save_root=None, nseq=None, L=None, layers=None, calib_batch=1, fused="exact",
                    reference_cadence=False, gguf=False):
    """Quantizer.quantize on a random-init Llama of the workload's architecture (the reference's timed region,
    quant.py:251-254) -> dict with wall seconds, Mparams/s and the split."""
    from gptq_gguf_toolkit_amd.quantizer import Quantizer
    cfg_kw = dict(wl["model"])
    if layers:
        cfg_kw["num_hidden_layers"] = layers
    nseq, L = nseq or wl["nseq"], L or wl["L"]
    t0 = time.perf_counter()
    model = build_model(cfg_kw, dev)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    g = torch.Generator().manual_seed(1)
    ids = [torch.randint(0, cfg_kw["vocab_size"], (1, L), generator=g) for _ in range(nseq)]
    ids = dist_utils.shard_calibration(ids, rank, world) if world > 1 else ids
    data = [([], {"input_ids": t}) for t in ids]
    # data.pth tree: 1.25 B/param.  tmpfs when it has room (the box's disk is an overlay), else the temp dir
    root = save_root
    if root is None:
        shm_free = shutil.disk_usage("/dev/shm").free if os.path.isdir("/dev/shm") else 0
        # (with the packer leg: + 2 B/param of checkpoint + 0.57 B/param of .gguf)
        root = "/dev/shm" if shm_free > (64e9 if gguf else 24e9) else tempfile.gettempdir()
    save_dir = tempfile.mkdtemp(prefix="gq_bench_", dir=root) if rank == 0 else None
    hf_dir = None
    if gguf and rank == 0 and world == 1:
        # the checkpoint directory the packer reads (the reference's `model` argument), written BEFORE the timed region from the
        # un-quantized weights: the quantizer writes the dequantized ones back into the live model
        t0 = time.perf_counter()
        try:
            need = 2.7 * sum(p.numel() for p in model.parameters())
            if shutil.disk_usage(root).free < 1.5 * need:
                raise OSError(f"{root}: not enough room for the checkpoint directory and the .gguf file")
            hf_dir = tempfile.mkdtemp(prefix="gq_bench_hf_", dir=root)
            model.save_pretrained(hf_dir, safe_serialization=True)
        except Exception as e:  # the quantizer leg must still run: the packer leg reports why it did not
            if hf_dir is not None:
                shutil.rmtree(hf_dir, ignore_errors=True)
            hf_dir, gguf_skip = None, repr(e)
        else:
            gguf_skip = None
        t_save = time.perf_counter() - t0
    if world > 1:
        box = [save_dir]
        dist.broadcast_object_list(box, src=0)
        save_dir = box[0]
    q = wl["q"]
    qc = {k: QT[q] for k in ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "down_proj", "up_proj",
                             "embed_tokens", "lm_head")}
    drv = Quantizer(model, data_loader=data, quantizable_modules=r".*layers.*((q|k|v|o|gate|up|down)_proj)$",
                    quantizer_kwargs=dict(QUANTIZER_KW, verbose=False), pre_block_modules=["model.embed_tokens"],
                    block_modules="model.layers", post_block_modules=["lm_head"], quant_non_block_modules=True,
                    device=str(dev), save_dir=save_dir, calibration_batch=calib_batch, fused_forward=fused,
                    # reference_cadence: two FULL forwards per block, as the reference runs it (quantizer.py:150-172);
                    # default: forward #1 stops at the last hooked Linear -- same saved bytes (tests/test_gpu_forward.py)
                    interrupt_forward1=not reference_cadence)
    params = sum(p.numel() for n, p in model.named_parameters() if p.dim() == 2)
    os.environ.setdefault("GQ_TIMING", "gpu")  # HIP-event split per phase next to the host-side one (read once, at the end)
    try:
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        # one untimed forward of one calibration sequence: the first use of every GEMM shape loads its hipBLASLt code object
        # (from a cold disk on a fresh box: +1.4 s in forward #1 of the first process, measured) -- warm-up, like the W steps
        if os.environ.get("GQ_BENCH_WM_NO_WARMUP") != "1":
            with torch.no_grad():
                model(input_ids=ids[0].to(dev), use_cache=False)
            torch.cuda.synchronize()
        # HIP-event time of the named kernels inside the run (default: the SYRK -- its roofline fraction on the activations of a
        # real forward, next to the synthetic step's; ~4 us per launch, 4 launches per block); e.g. "syrk,transpose16"
        wm_prof = os.environ.get("GQ_BENCH_WM_PROF", "syrk")
        if wm_prof:
            _cabi.prof_enable(wm_prof.split(","))
        t1 = time.perf_counter()
        drv.quantize(qc)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t1
        if wm_prof:
            prof_got = {k: (round(v[0], 1), v[1]) for k, v in _cabi.prof_collect().items() if v[1]}
            _cabi.prof_enable([])
        tm = torch.tensor([wall], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        wall = float(tm.item())
        files = sum(len(f) for _, _, f in os.walk(save_dir)) if rank == 0 else 0
        nbytes = sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(save_dir) for f in fs) if rank == 0 else 0
        e2e = {"error": gguf_skip} if (gguf and rank == 0 and world == 1 and hf_dir is None) else None
        if hf_dir is not None:
            try:
                e2e = gguf_leg(hf_dir, save_dir, root, wall)
                e2e["checkpoint_save_s_untimed"] = round(t_save, 1)
            except Exception as e:  # the bench line must still print
                e2e = {"error": repr(e)}
    finally:
        if rank == 0:
            shutil.rmtree(save_dir, ignore_errors=True)
            if hf_dir is not None:
                shutil.rmtree(hf_dir, ignore_errors=True)
    syrk_wm = None
    if wm_prof and "syrk" in prof_got and prof_got["syrk"][0] > 0:
        # algorithmic flops of the run's Hessian folds: per block the four distinct inputs (q/k/v, o, gate/up: hidden wide; down:
        # intermediate wide), T C (C + 128) each, T = this rank's calibration tokens
        hdim, idim, nblk = cfg_kw["hidden_size"], cfg_kw["intermediate_size"], cfg_kw["num_hidden_layers"]
        T_ = float(len(ids) * L)
        fl = nblk * T_ * (3.0 * hdim * (hdim + 128) + idim * (idim + 128.0))
        a_ = fl / (prof_got["syrk"][0] * 1e-3) / 1e12
        syrk_wm = {"achieved": round(a_, 1), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(a_ / PEAK_F16_MFMA_TFLOPS, 4),
                   "launches": prof_got["syrk"][1], "ms": prof_got["syrk"][0],
                   "operands": "bf16 activations of the model's own forward (random-init weights): sum of launch durations"}
    out = {"model": f"random-init LlamaForCausalLM {cfg_kw['num_hidden_layers']} layers, hidden {cfg_kw['hidden_size']}, "
                    f"bf16, attn {os.environ.get('GQ_ATTN', 'sdpa')}; embed + lm_head RTN ({q}), all block Linears GPTQ ({q})",
           "forward": {"off": "HF eager modules (the reference's forward)",
                       "exact": "HF modules with the bit-exact HIP kernels (rotary embedding, SwiGLU, RMSNorm in ATen's "
                                "summation order, verified at run time): same saved bytes as HF eager",
                       "all": "as exact; RMSNorm through the free-order kernel (<= 2 ulp) where the ordered one is not verified"}[drv.fused_forward],
           "block_forwards": ("two full forwards per block (reference quantizer.py:150-172)" if reference_cadence else
                              "forward #1 stops at the last hooked Linear (the reference discards its output too); forward #2 in full"),
           "calib": f"{nseq}x{L} synthetic ids ({len(ids)} sequences on this rank), {calib_batch} per block forward"
                    + (" (the reference's cadence)" if calib_batch == 1 else " (--calibration_batch: same Hessian sums, fewer and larger GEMMs)"),
           "params_quantized_M": round(params / 1e6, 1),
           "wall_s_quantizer_region": round(wall, 2), "Mparams_per_s": round(params / wall / 1e6, 1),
           "syrk_roofline_on_model_activations": syrk_wm,
           "split": dict(drv.timing, **({"kernel_ms_launches": prof_got} if wm_prof else {})), "schedule": getattr(drv, "schedule_stats", None), "model_build_s": round(t_build, 1),
           "data_pth": {"files": files, "GB": round(nbytes / 1e9, 2), "dir": root},
           "end_to_end_gguf": e2e,
           "region": "Quantizer.quantize (reference quant.py:251-254), model and ids resident on the GPU/host before it, "
                     "after one untimed forward of one sequence (GEMM code objects loaded)"}
    del drv, model
    torch.cuda.empty_cache()
    return out


# ----------------------------------------------------------------------------- main
def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: re-exec under torch.distributed.run with one rank per GPU,
    the way the reference's run_quant.sh:15 starts its own ranks (torchrun --nnodes=1 --nproc-per-node=$NUM_GPUS) and
    quant.py:149-155 joins them.  nccl (= RCCL) needs N visible GPUs; fewer is refused unless --backend gloo was asked for."""
    import socket
    n_dev = torch.cuda.device_count()
    if args.backend != "gloo" and n_dev < args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but {n_dev} GPU(s) visible: RCCL needs one GPU per rank "
                 f"(pass --backend gloo to run {args.gpus} ranks that share the visible GPUs: a code-path check, not a measurement)")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    argv = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.stdout.flush()
    sys.stderr.flush()
    os.execvpe(sys.executable, argv, env)


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def compact_line(line):
    """The headline as ONE short JSON line (<= 2 KB, scalars only inside `roofline` / `cpu_baseline` / `config`): the
    last line of stdout.  Everything else the run measured is on the line before it and in
    gpurun_out/bench_detail_<workload>_n<N>.json."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "step_latency_ms", "ranks_seen", "collective_backend")
    out = {k: line.get(k) for k in keep if k in line}
    cfg = line.get("config", {})
    out["config"] = {"workload": str(cfg.get("workload", ""))[:96], "parallelism": cfg.get("parallelism"),
                     "calib_seqs_per_rank": cfg.get("calib_seqs_per_rank")}
    r = dict(line.get("roofline") or {})
    t = r.pop("traffic", None)
    if isinstance(r.get("kernel"), str):
        r["kernel"] = r["kernel"][:64]
    if isinstance(t, dict) and t.get("GB_per_launch"):
        # HBM-side bytes per launch from the separate --pmc pass (profiles/r06_syrk_traffic.json), GB
        r["traffic"] = t["GB_per_launch"]
        r["traffic_unit"] = "GB/launch L2-miss reads"
        if t.get("algorithmic_GB_per_launch"):
            r["traffic_algorithmic"] = t["algorithmic_GB_per_launch"]
            r["traffic_ratio"] = round(t["GB_per_launch"] / t["algorithmic_GB_per_launch"], 2)
    else:
        r["traffic"] = None
    tu = line.get("trailing_update") or {}
    for k_out, path in (("trailing_far_alone_frac", ("far_alone", "frac")), ("trailing_whole_alone_frac", ("whole_alone", "frac")),
                        ("trailing_far_in_region_frac", ("far_in_region", "frac")),
                        ("trailing_far_in_region_over_busy", ("far_in_region", "frac_over_busy_time")),
                        ("trailing_loop_ms_as_run", ("loop_ms", "as_run"))):
        r[k_out] = _get(tu, *path)
    r["trailing_peak_f32"] = PEAK_F32_MFMA_TFLOPS
    r["frac_on_model_forward_activations"] = _get(line, "whole_model", "syrk_roofline_on_model_activations", "frac")
    r["column_loop_ns_per_step"] = _get(line, "column_loop", "ns_per_step")
    r["encoders_frac_of_hbm"] = _get(line, "encoders", "frac")
    out["roofline"] = r
    c = dict(line.get("cpu_baseline") or {})
    if c:
        st = c.pop("stages_s_per_block", None) or {}
        so = c.pop("step_only", None) or {}
        c["sample"] = str(c.get("sample", ""))[:60]
        c.update({f"stage_{k}": v for k, v in st.items()})
        c["step_only_value"] = so.get("value")
    out["cpu_baseline"] = c or None  # (None at N > 1: the baseline is part of the N = 1 line)
    for key in ("whole_model", "whole_model_hf_eager", "whole_model_batch4"):
        out[f"{key}_wall_s"] = _get(line, key, "wall_s_quantizer_region")
    # quantize -> pack_gptq_into_gguf -> .gguf (the north star's end-to-end target), N = 1
    out["end_to_end_gguf_wall_s"] = _get(line, "whole_model", "end_to_end_gguf", "end_to_end_gguf_wall_s")
    out["gguf_pack_wall_s"] = _get(line, "whole_model", "end_to_end_gguf", "pack_wall_s")
    sp = _get(line, "whole_model", "end_to_end_gguf", "split_s") or {}
    out["gguf_pack_split_s"] = ("load %s h2d %s pack %s d2h %s write %s wait %s" % tuple(sp.get(k) for k in ("load", "h2d", "permute_pack", "d2h", "write", "wait"))) if sp else None
    out["collectives_per_step"] = line.get("collectives_per_step")
    out["allreduce_probe_ms"] = _get(line, "allreduce_probe", "ms")
    rnd = lambda v: round(v, 5) if isinstance(v, float) else v  # noqa: E731
    out["tolerance_ints_differ"] = rnd(_get(line, "tolerance_parity", "ints_differ"))
    out["tolerance_noise_floor"] = rnd(_get(line, "tolerance_parity", "ulp_noise_floor", "ints_differ"))
    out["detail"] = "previous stdout line"
    return out


def emit(line, args, world):
    """Two stdout lines on rank 0: the full dict (everything measured), then the compact headline LAST."""
    full = json.dumps(line)
    print(full)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", f"bench_detail_{args.workload}_n{world}.json"), "w") as f:
            f.write(full + "\n")
    except OSError:
        pass
    print(json.dumps(compact_line(line)))
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="llama3-8b-block-q4k", choices=list(WORKLOADS))
    ap.add_argument("--calib-seqs", type=int, default=None)
    ap.add_argument("--seq-len", type=int, default=None)
    ap.add_argument("--layers", type=int, default=None, help="whole-model workloads: number of blocks (default: all)")
    ap.add_argument("--calib-batch", type=int, default=1, help="whole-model workload: calibration samples per block forward")
    ap.add_argument("--fused-forward", nargs="?", const="all", default="exact", choices=["off", "exact", "all"],
                    help="whole-model workload: forward kernels (exact = the Quantizer's default; bare flag = all)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-whole-model", action="store_true", help="skip the whole-model leg of the block workloads")
    ap.add_argument("--no-side-legs", action="store_true", help="skip trailing_update / tolerance_parity legs")
    ap.add_argument("--breakdown", action="store_true", help="one extra profiled step: per-kernel ms to stderr")
    ap.add_argument("--backend", default=None, choices=["nccl", "gloo"],
                    help="torch.distributed backend: nccl (= RCCL over xGMI, default); gloo only to exercise the "
                         "N>1 code path with several ranks sharing one GPU (must be asked for explicitly)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)  # does not return: `python bench.py --gpus N` starts its N ranks itself (run_quant.sh:15)
    args.backend = args.backend or "nccl"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP extension has no CPU fallback)"
    local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ranks_seen, ar_probe = 1, None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group("gloo")
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)  # proves the collective backend is alive: every rank contributed
        ranks_seen = int(one.item())
        assert ranks_seen == dist.get_world_size() == world
    # a rank mismatch is fatal: a line that says n_gpus = 1 under --gpus 8 is not a measurement of 8 GPUs
    assert ranks_seen == world == args.gpus, f"--gpus {args.gpus} but {ranks_seen} rank(s) answered (WORLD_SIZE={world})"
    if world > 1 and args.backend == "nccl":
        assert torch.cuda.device_count() >= world, (f"--backend nccl needs one GPU per rank: {torch.cuda.device_count()} "
                                                    f"visible, {world} ranks")
    wl = WORKLOADS[args.workload]
    nseq, L = args.calib_seqs or wl["nseq"], args.seq_len or wl["L"]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if "model" in wl:  # a step = the whole model through Quantizer.quantize
        runs = []
        for i in range(args.warmup + args.steps):
            runs.append(whole_model_run(wl, dev, world, rank, nseq=nseq, L=L, layers=args.layers, calib_batch=args.calib_batch, fused=args.fused_forward))
        timed = runs[args.warmup:]
        dt = sum(r["wall_s_quantizer_region"] for r in timed)
        if rank == 0:
            params = timed[-1]["params_quantized_M"] * 1e6
            emit({
                "metric": "Mparams/s GPTQ-quantized", "value": round(params * args.steps / dt / 1e6, 2),
                "unit": "Mparams/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(dt / args.steps * 1e3, 1), "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"{args.workload}: Quantizer.quantize of the whole model, {nseq}x{L}-token "
                                       f"calibration", "parallelism": f"calib-dp{world}+matrix-fanout"},
                "ranks_seen": ranks_seen, "whole_model": timed[-1]}, args, world)
        if world > 1:
            dist.destroy_process_group()
        return

    shapes = wl["shapes"]
    nseq_local = nseq // world  # contiguous shard, remainder dropped (quant.py:177-179)
    params = sum(R * C for R, C, _ in shapes.values())
    layers, W16 = make_layers(shapes, dev)
    X = make_inputs(wl, nseq_local, L, dev, seed=1 + rank)
    torch.cuda.synchronize()

    for _ in range(args.warmup):
        block_step(wl, layers, W16, X)
    sync()
    if world > 1:
        # all-reduce of the widest Hessian's payload, alone: bytes per rank and time (RCCL over xGMI)
        C = max(c for _, c, _ in shapes.values())
        H = torch.zeros(C, C, device=dev)
        dist_utils.allreduce_hessian(H)
        sync()
        t0 = time.perf_counter()
        dist_utils.allreduce_hessian(H)
        torch.cuda.synchronize()
        ar_probe = {"C": C, "payload_MB": round(dist_utils.hessian_payload_bytes(C) / 1e6, 1),
                    "ms": round((time.perf_counter() - t0) * 1e3, 3), "backend": args.backend}
        del H
    if world > 1 and os.environ.get("GQ_BENCH_VERIFY") == "1":  # every rank must hold the same results
        block_step(wl, layers, W16, X)
        sync()
        for name in sorted(layers):
            w = layers[name].weight.data
            cs_ = w.double().sum().reshape(1)
            lo_, hi_ = cs_.clone(), cs_.clone()
            dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
            assert float(lo_) == float(hi_) and float(w.float().abs().sum()) > 0, f"{name}: ranks disagree"
        if rank == 0:
            print("verify: all ranks hold identical results", file=sys.stderr)
    # the dominant kernel (fp16 MFMA SYRK) and the far trailing update are timed live with HIP events on their
    # launch streams, inside the timed region
    _cabi.prof_enable(["syrk", "trailing_far_gemm32"])
    keep = {}
    sched = None
    coll0, collb0 = dict(dist_utils.collective_calls), dict(dist_utils.collective_bytes)
    t0 = time.perf_counter()
    for i in range(args.steps):
        sched = block_step(wl, layers, W16, X, keep=keep if i == args.steps - 1 else None)
    sync()
    dt = time.perf_counter() - t0
    prof = _cabi.prof_collect(busy=True)
    _cabi.prof_enable([])
    # data-path collectives one step issued on this rank (N > 1: one all-reduce per distinct Hessian, ONE all-gather of
    # the block's results, no broadcast -- asserted by tests/test_gpu_round3.py)
    coll = {k: (v - coll0.get(k, 0)) / args.steps for k, v in dist_utils.collective_calls.items()}
    coll_bytes = {k: (v - collb0.get(k, 0)) / args.steps for k, v in dist_utils.collective_bytes.items()}
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    BlockSchedule.verify()  # the device flags of every reused factorisation of the timed steps
    # latency of ONE step alone (host and device idle before and after): the scheduler never synchronises, so the
    # K timed steps above are enqueued back to back and the tail of a step's chains (latency-bound, few CUs) runs
    # under the next step's SYRK; inside a model the block's second forward takes that place
    sync()
    t1 = time.perf_counter()
    block_step(wl, layers, W16, X)
    sync()
    latency_ms = (time.perf_counter() - t1) * 1e3
    BlockSchedule.verify()

    if args.breakdown and rank == 0:
        _cabi.prof_enable(None)
        block_step(wl, layers, W16, X)
        torch.cuda.synchronize()
        bd = _cabi.prof_collect(busy=True)  # GQ_PROF_DUMP=<file> also writes the interval timeline
        _cabi.prof_enable([])
        tot = sum(v[0] for v in bd.values())
        for k, (ms, n, busy) in sorted(bd.items(), key=lambda kv: -kv[1][0]):
            print(f"  {k:22s} {ms:10.2f} ms  {n:6d} launches  {100 * ms / tot:5.1f} %  busy {busy:8.2f} ms", file=sys.stderr)

    if rank == 0:
        # roofline of the dominant kernel: algorithmic MFMA flops of the upper-triangular SYRK (T * C * (C + 128) per
        # distinct input) over the live HIP-event time: busy_ms = the UNION of the launch intervals;
        # sum_launch_ms / launches is the plain per-launch average rocprof reports
        syrk_sum_ms, syrk_n, syrk_ms = prof.get("syrk", (0.0, 0, 0.0))
        flops = syrk_flops(wl, X) * args.steps
        ach = flops / (syrk_ms * 1e-3) / 1e12 if syrk_ms > 0 else None
        # roofline.traffic: L2-miss reads per SYRK launch from a SEPARATE rocprofv3 --pmc pass of this command (the
        # driver's run cannot carry counters: a --pmc run serialises the kernels).  The file names the kernel sources it
        # was measured on; for any other build the field is null -- a stale constant is not a measurement.
        traffic = None
        # (the newest profiles/r*_syrk_traffic.json; it names the kernel sources it was measured on)
        import glob
        import hashlib
        tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_syrk_traffic.json")))
        if tfiles and args.workload == "llama3-8b-block-q4k" and world == 1 and not args.calib_seqs and not args.seq_len:
            tpath = tfiles[-1]
            try:
                tj = json.load(open(tpath))  # written by profiles/collect_r06.sh
                hsh = hashlib.sha256()
                for fn in sorted(glob.glob(os.path.join(ROOT, "gptq-gguf-toolkit_amd", "csrc", "*.h*"))):
                    hsh.update(open(fn, "rb").read())
                if tj.get("kernel_sources_sha256") == hsh.hexdigest()[:16]:
                    traffic = {"GB_per_launch": tj["GB_per_launch"], "algorithmic_GB_per_launch": tj.get("algorithmic_GB_per_launch"),
                               "measured_on": tj.get("measured_on"), "launches": tj.get("launches")}
                else:
                    traffic = {"GB_per_launch": None, "note": f"profiles/{os.path.basename(tpath)} was measured on other kernel "
                                                              "sources than this build's: not reported"}
            except Exception:
                traffic = None
        roof = {"bound": "mfma",
                "kernel": ("syrk16_256w_kernel<f16>: 4 waves, 128x128 wave tiles" if ops.option_get("syrk_w4")
                           else "syrk16_256n_kernel<f16>: 8 waves, 128x64 wave tiles") + " (gq_h_accumulate_grouped)",
                "achieved": round(ach, 2) if ach else None, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_F16_MFMA_TFLOPS, 4) if ach else None,
                "traffic": traffic,
                "launches": syrk_n, "busy_ms_per_step": round(syrk_ms / args.steps, 3),
                "avg_launch_ms": round(syrk_sum_ms / max(syrk_n, 1), 4),
                "share_of_step": round(syrk_ms / 1e3 / dt, 3)}
        far = prof.get("trailing_far_gemm32", (0.0, 0, 0.0))
        # the side legs are single-GPU measurements of the N = 1 line; some of them drive handles whose quantize() issues
        # collectives, which only rank 0 would enter here (a hang under RCCL): never at N > 1
        side = not args.no_side_legs and world == 1
        line = {
            "metric": "Mparams/s GPTQ-quantized", "value": round(params * args.steps / dt / 1e6, 2),
            "unit": "Mparams/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "step_latency_ms": round(latency_ms, 2),
            "config": {"workload": f"{args.workload}: {len(shapes)} Linears of one block ({params / 1e6:.1f} M params), "
                                   f"{nseq}x{L}-token calibration fed per sequence through the package's BlockSchedule, "
                                   f"block_size 128, rel_damp 0.01, nstep 20",
                       "calib_seqs_per_rank": nseq_local, "parallelism": f"calib-dp{world}+matrix-fanout",
                       "owners": sched.owners if world > 1 else "rank0"},
            "ranks_seen": ranks_seen, "collective_backend": (f"{args.backend} (RCCL over xGMI)" if args.backend == "nccl"
                                                             else args.backend) if world > 1 else None,
            "collectives_per_step": coll if world > 1 else None,
            "collective_bytes_per_step": coll_bytes if world > 1 else None,  # this rank's payloads
            "allreduce_probe": ar_probe, "schedule": sched.stats,
            "wall_s_llama3_8b_32_blocks_extrapolated": round(dt / args.steps * 32, 2)
            if args.workload.startswith("llama3-8b") else None,
            "roofline": roof,
            "trailing_update": (tu := trailing_update_legs(wl, W16, X, (far[0], far[1], far[2], args.steps)) if side else None),
            "encoders": (el := encoder_legs(wl, layers, W16, X, tu) if side else (None, None))[0],
            "column_loop": el[1],
            "tolerance_parity": tolerance_parity(wl, W16, X) if side else None,
            "tolerance_parity_widest": tolerance_parity(wl, W16, X, widest=True, n_rows=128) if side else None,
            "fast_obq": fast_obq_leg(wl, W16, X) if side else None,
            # the host-core baseline belongs to the N = 1 line (rank 0 holds every U there; at N > 1 the ranks' host cores are
            # busy feeding their GPUs and rank 0 factorised only the matrices it owns)
            "cpu_baseline": None if (args.no_cpu_baseline or world > 1) else cpu_baseline(wl, W16, keep),
        }
    del keep, sched
    if args.workload.startswith("llama3-8b-block") and not args.no_whole_model:
        # the drop-in pipeline end to end (all ranks take part: calibration shards, collectives)
        del layers, W16, X
        torch.cuda.empty_cache()
        wm = {}
        for key, cb, fused in (("whole_model", 1, "exact"), ("whole_model_hf_eager", 1, "off"), ("whole_model_batch4", 4, "exact")):
            try:
                wm[key] = whole_model_run(WORKLOADS["llama3-8b-model-q4k"], dev, world, rank, layers=args.layers, calib_batch=cb,
                                          fused=fused, reference_cadence=(fused == "off"), gguf=(key == "whole_model"))
            except Exception as e:  # the bench line must still print
                wm[key] = {"error": repr(e)}
        if rank == 0:
            line.update(wm)
    if rank == 0:
        emit(line, args, world)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
