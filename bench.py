#!/usr/bin/env python3
"""bench.py -- Mparams/s GPTQ-quantized on synthetic Llama-shaped Linears (BASELINE.json metric).

One "step" = one pass of the hot path over ONE Llama-3-8B transformer block
(configs[1]: Llama-3-8B -> Q4_K, 128 x 2048-token calibration): 218.1 M parameters in 7
Linears.  Inside the timed region, per step:
  * H accumulation from the calibration activations of the 4 distinct Linear inputs
    (attn-in feeds q/k/v, o-in, mlp-in feeds gate/up, down-in), one gq_h_accumulate per
    sequence exactly like the reference's forward hook (gptq.py:79-114);
  * [N>1] one RCCL all-reduce (AVG) per distinct Hessian (gptq.py:131-132);
  * per Linear, on its owner rank: fp32 working copy, gq_h_prepare (damping + Cholesky
    chain), gq_gptq_quantize (scale search + column loop + trailing update),
    gq_dequantize to fp16 (the write-back of quantizer.py:257-264) and gq_pack
    (GGUF block bytes);
  * [N>1] broadcast of the dequantized fp16 weight from the owner (needed by every rank
    for the block's second forward).
Inputs (weights, activations) are resident in HBM before the timed region starts.
There is no model forward here (synthetic Linears), so this is the GPTQ.quantize region
of quantizer.py:248-265 plus the hook-side H updates.

Prints ONE JSON line on rank 0 (see the driver contract).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "8")  # cpu_baseline threads (the reference's run_quant.sh:9 pins 8 too)
# One hardware queue per HIP stream: with the runtime's default of 4 two of the block's independent chains share a
# queue and run one after the other (kernel trace: same Queue_Id).  Must be set before the HIP runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from gptq_gguf_toolkit_amd import _cabi, dist_utils, ops  # noqa: E402

Q4_K = 12
# L2-miss read bytes per SYRK launch of THIS command (rocprofv3 --pmc FETCH_SIZE x 2 KB, profiles/pmc_bench_fetch.sh,
# profiles/r01_syrk_pmc.txt), by sequences per launch: 32 -> (4 x 7.9 + 4 x 33.7) / 8, 64 -> (2 x 15.7 + 2 x 60.0) / 4
SYRK_TRAFFIC_GB_PER_LAUNCH = {32: 20.68, 64: 37.67}
# Llama-3-8B block: name -> (R, C, input group)
LLAMA3_8B = {
    "q_proj": (4096, 4096, "attn_in"), "k_proj": (1024, 4096, "attn_in"), "v_proj": (1024, 4096, "attn_in"),
    "o_proj": (4096, 4096, "o_in"), "gate_proj": (14336, 4096, "mlp_in"), "up_proj": (14336, 4096, "mlp_in"),
    "down_proj": (4096, 14336, "down_in"),
}
TINY = {  # TinyLlama-1.1B block (configs[0] shapes) for quick runs
    "q_proj": (2048, 2048, "attn_in"), "k_proj": (256, 2048, "attn_in"), "v_proj": (256, 2048, "attn_in"),
    "o_proj": (2048, 2048, "o_in"), "gate_proj": (5632, 2048, "mlp_in"), "up_proj": (5632, 2048, "mlp_in"),
    "down_proj": (2048, 5632, "down_in"),
}
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X dense fp16/bf16 MFMA (MI355X_MICROARCH.md)
ROW_SPLIT = -1  # owner id of a matrix every rank quantizes on its own rows (dist_utils.row_split_names)
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X fp32 matrix (v_mfma_f32_32x32x2_f32: 256 flop/clk/CU x 256 CUs x 2.4 GHz)


def make_inputs(shapes, nseq, L, dev, seed=1):
    """X ~ N(0,1) * sigma_c, sigma_c log-normal, 0.1 % outlier channels x20, fp16."""
    g = torch.Generator(device=dev).manual_seed(seed)
    X = {}
    for name, (R, C, inp) in shapes.items():
        if inp in X:
            continue
        sig = torch.exp(torch.randn(C, device=dev, generator=g) * 0.5)
        nout = max(1, C // 1000)
        sig[torch.randperm(C, device=dev, generator=g)[:nout]] *= 20.0
        x = torch.empty(nseq, L, C, device=dev, dtype=torch.float16)
        for s in range(nseq):
            x[s] = (torch.randn(L, C, device=dev, generator=g) * sig).half()
        X[inp] = x
    return X


def make_weights(shapes, dev, seed=0):
    W = {}
    for i, (name, (R, C, _)) in enumerate(shapes.items()):
        g = torch.Generator(device=dev).manual_seed(seed + i)
        W[name] = (torch.randn(R, C, device=dev, generator=g) * 0.02).half()
    return W


def quantize_block(shapes, W16, X, owners, rank, world, q_type=Q4_K, block_size=128, rel_damp=0.01, keep=None,
                   hbatch=None, hws=None, streams=None, row_chunks=1):
    dev = next(iter(W16.values())).device
    # ---- Hessians: one per distinct input.  The activations of `hbatch` sequences are folded in
    # per launch (beta = n/(n+b), alpha = 2/(n+b): the telescoped form of b single-sample updates of
    # gptq.py:106-112), all distinct inputs of the block in ONE grouped SYRK grid.
    H = {inp: torch.zeros(x.shape[-1], x.shape[-1], device=dev, dtype=torch.float32) for inp, x in X.items()}
    # Two grouped grids, ONE AFTER THE OTHER on the main stream: the narrow inputs first (12 ms), then the widest
    # one (43 ms) alone on the chip; then the four chains (prepare -> column loop) on their streams.
    # Measured against the widest-first / concurrent-grids schedule: 105.4 vs 107.3 ms per step on one box, equal
    # on another; never worse, and no side stream is needed.
    names = sorted(X, key=lambda i: -X[i].shape[-1])
    splits = [names[1:], names[:1]] if len(names) > 1 and world == 1 else [names]
    if os.environ.get("GQ_BENCH_CONCURRENT_GRIDS") and len(splits) == 2:  # A/B: the previous schedule
        splits = [splits[1], splits[0]]
    nseq = X[names[0]].shape[0]
    hb = hbatch or nseq
    ev_ready = {}
    main = torch.cuda.current_stream(dev)
    # The second grid runs on a side stream with its own slice of the workspace: its workgroups fill the
    # CUs the first grid's last (partial) round of tiles leaves idle.
    side = streams[-1] if (streams and len(splits) > 1 and os.environ.get("GQ_BENCH_CONCURRENT_GRIDS")) else None
    if side is not None:
        ev0 = torch.cuda.Event()
        ev0.record(main)
        side.wait_event(ev0)
    off = 0
    for k, grp in enumerate(splits):
        st = side if (k > 0 and side is not None) else main
        need = sum(ops.workspace_bytes(_cabi.WS_H_ACCUMULATE, 0, X[i].shape[-1], hb * X[i].shape[1]) for i in grp)
        wsk = hws[off:off + need] if hws is not None else None
        off += need
        with torch.cuda.stream(st):
            n = 0
            while n < nseq:
                b = min(hb, nseq - n)
                ops.h_accumulate_grouped([H[i] for i in grp], [X[i][n:n + b].reshape(-1, X[i].shape[-1]) for i in grp],
                                         [n / (n + b)] * len(grp), [2.0 / (n + b)] * len(grp), ws=wsk)
                n += b
        ev = torch.cuda.Event()
        ev.record(st)
        for i in grp:
            ev_ready[i] = ev
    if world == 1 and not os.environ.get("GQ_BENCH_CONCURRENT_GRIDS"):
        # every chain starts after BOTH grids: a resident SYRK grid (one 128 KiB-LDS, 8-wave workgroup per CU) leaves
        # the chains of the narrow inputs nothing but the gaps between its tiles anyway, and without them in its
        # way the SYRK sustains 1.24 instead of 1.22 PFLOP/s (step 104.5 vs 104.9 ms on the same box)
        for i in ev_ready:
            ev_ready[i] = ev
    if world > 1:  # widest first: its chain is the critical one and starts as soon as ITS Hessian is reduced
        for inp in names:
            dist_utils.allreduce_hessian(H[inp])  # RCCL over xGMI, upper-triangular tiles only
            ev = torch.cuda.Event()
            ev.record(main)
            ev_ready[inp] = ev
    # ---- per Linear on its owner.  Linears fed by the same input share H, hence U when their
    # dead/zero-column sets agree (gq_w_prepare checks; the leader's U is then bit-identical).
    # The input groups are independent chains (prepare -> column loop), so each runs on its own HIP
    # stream: the single-workgroup diagonal factorisations and the 64-wave column-loop kernels of one
    # chain overlap with the GEMMs of the others.
    out, pending = {}, []
    groups = {}
    split = {n for n, o in owners.items() if o == ROW_SPLIT}  # every rank: factorise, quantize its own rows
    for name, (R, C, inp) in shapes.items():
        if owners[name] == rank or name in split:
            groups.setdefault(inp, []).append(name)
    # the critical chain first: row-split matrices (their factorisation is replicated on every rank), then by cost
    order = sorted(groups, key=lambda g: (not any(n in split for n in groups[g]),
                                          -sum(shapes[n][0] * shapes[n][1] * shapes[n][1] for n in groups[g])))
    host_trace = os.environ.get("GQ_BENCH_TRACE_HOST")
    t_host0 = time.perf_counter()
    for gi, inp in enumerate(order):
        if host_trace:
            print(f"  [host] +{1e3 * (time.perf_counter() - t_host0):7.2f} ms: enqueue chain {gi} ({inp}: {groups[inp]})", file=sys.stderr)
        st = streams[gi % len(streams)] if streams else main
        st.wait_event(ev_ready[inp])
        with torch.cuda.stream(st):
            U = flag = cf = None
            for name in groups[inp]:
                R, C, _ = shapes[name]
                Wf = W16[name].float()
                mm = None
                if U is None:
                    Hc = H[inp].clone()  # each reference handle damps its own H
                    U, flag, cf = ops.h_prepare(Hc, Wf, rel_damp, want_flags=True)
                    del Hc
                else:
                    mm = ops.w_prepare(cf, Wf)  # speculative reuse of the leader's U, verified below
                # the widest Linear is the critical chain: its rows are split over side streams
                chunks = row_chunks if (gi == 0 and len(groups[inp]) == 1) else 1
                Wq = Wf
                if name in split:
                    r0, r1, rchunk = dist_utils.row_slice(R, rank, world)
                    Wq = Wf[r0:r1]
                if Wq.shape[0] > 0:
                    q, d, s, dmin, m = ops.gptq_quantize(Wq, U, q_type, block_size, row_chunks=chunks)
                    deq = ops.dequantize(q_type, q, d, s, dmin, m, torch.float16)
                    packed = ops.pack(q_type, q, d, s, dmin, m)
                else:
                    deq = torch.empty(0, C, device=dev, dtype=torch.float16)
                    q = d = s = dmin = m = packed = None
                out[name] = deq
                pending.append((name, inp, mm))
                if keep is not None:
                    keep[name] = (q, d, s, dmin, m, packed, flag, U if name == "k_proj" else None)
                del Wf
            del U
        ev = torch.cuda.Event()
        ev.record(st)
        main.wait_event(ev)
    for name, inp, mm in pending:  # a follower whose zero-column set differs gets its own factorisation
        if mm is not None and int(mm.item()) != 0:
            R, C, _ = shapes[name]
            Wf = W16[name].float()
            U, flag = ops.h_prepare(H[inp].clone(), Wf, rel_damp)
            q, d, s, dmin, m = ops.gptq_quantize(Wf, U, q_type, block_size)
            out[name] = ops.dequantize(q_type, q, d, s, dmin, m, torch.float16)
            ops.pack(q_type, q, d, s, dmin, m)
    if world > 1:
        for name, (R, C, inp) in shapes.items():
            if name in split:  # all-gather of the ranks' row slices (padded to the common chunk height)
                r0, r1, rchunk = dist_utils.row_slice(R, rank, world)
                out[name] = dist_utils.all_gather_rows(out[name], R, rchunk)
                continue
            if name not in out:
                out[name] = torch.empty(R, C, device=dev, dtype=torch.float16)
            dist.broadcast(out[name], src=owners[name])
    return out


def trailing_update_roofline(shapes, W16, q_type=Q4_K, block_size=128):
    """The blocked trailing-update GEMM (north star: >= 70 % of the fp32 MFMA peak) of the widest Linear, alone on
    the GPU after the timed region: the chained far updates of one gq_gptq_quantize, HIP events on the launch
    stream.  Algorithmic flops of one launch = 2 * R * 1024 * (C - S1) (8 look-ahead blocks of 128 columns)."""
    try:
        name = max(shapes, key=lambda n: shapes[n][1] * shapes[n][0] * shapes[n][1])
        R, C, _ = shapes[name]
        dev = W16[name].device
        U = torch.eye(C, device=dev) + torch.triu(torch.randn(C, C, device=dev) * 0.01, 1)
        sb = 8 * block_size
        flops = sum(2.0 * R * (min(s0 + sb, C) - s0) * (C - min(s0 + sb, C)) for s0 in range(0, C, sb))
        best = None
        for it in range(2):
            Wf = W16[name].float()
            torch.cuda.synchronize()
            _cabi.prof_enable(["trailing_far_gemm32"])
            ops.gptq_quantize(Wf, U, q_type, block_size)
            torch.cuda.synchronize()
            ms, n, _ = _cabi.prof_collect(busy=True).get("trailing_far_gemm32", (0.0, 0, 0.0))
            _cabi.prof_enable([])
            if n and (best is None or ms < best[0]):
                best = (ms, n)
        if not best:
            return None
        ach = flops / (best[0] * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": "gemm32_chain_full_kernel<128> (far trailing update of gq_gptq_quantize)",
                "linear": f"{name} {R}x{C}", "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "launches": best[1],
                "avg_launch_ms": round(best[0] / best[1], 4), "measured": "alone on the GPU, after the timed region"}
    except Exception as e:  # the bench line must still print
        return {"error": repr(e)}


def cpu_baseline(shapes, W16, keep):
    """Oracle (C restatement of the reference, OpenMP) on the host cores: GPTQ.step of the three Linears fed by
    the attention input (q/k/v: 25.2 M params, one shared U -- the U the GPU used).  ~10-30 s of CPU work."""
    try:
        from oracle import oracle as O
        U = keep["k_proj"][7].cpu().numpy()
        threads = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
        names = [n for n in ("q_proj", "k_proj", "v_proj") if n in shapes]
        tot, same, cnt, dt = 0, 0.0, 0, 0.0
        for n in names:
            R, C, _ = shapes[n]
            W = W16[n].float().cpu().numpy()
            t0 = time.perf_counter()
            _, oq, *_ = O.gptq_step(W, U, Q4_K, block_size=128)
            dt += time.perf_counter() - t0
            tot += R * C
            same += float((oq == keep[n][0].cpu().numpy()).sum())
            cnt += oq.size
        return {"value": round(tot / dt / 1e6, 3), "unit": "Mparams/s", "cores": threads, "kind": "port",
                "sample": f"GPTQ.step (scale search + column loop + trailing update, given U) of the {len(names)} "
                          f"Q4_K Linears fed by the attention input ({'/'.join(names)}, {tot / 1e6:.1f} M params), "
                          f"{dt:.1f} s; ints equal to the GPU's: {same / cnt:.6f}"}
    except Exception as e:  # the bench line must still print
        return {"value": None, "unit": "Mparams/s", "cores": 0, "kind": "port", "sample": f"failed: {e!r}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="llama3-8b-block-q4k", choices=["llama3-8b-block-q4k", "tinyllama-block-q4k"])
    ap.add_argument("--calib-seqs", type=int, default=None)
    ap.add_argument("--seq-len", type=int, default=None)
    ap.add_argument("--hessian-batch", type=int, default=None,
                    help="sequences folded into H per SYRK launch (default: 32; 1 = reference cadence)")
    ap.add_argument("--streams", type=int, default=4, help="HIP streams for the independent per-input chains (0: one)")
    ap.add_argument("--row-chunks", type=int, default=1,
                    help="row chunks (side streams) for the column loop of the block's widest Linear")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="one extra profiled step: per-kernel ms to stderr")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend: nccl (= RCCL over xGMI, default); gloo only to exercise the "
                         "N>1 code path with several ranks sharing one GPU")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP extension has no CPU fallback)"
    local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group("gloo")
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    if args.workload == "llama3-8b-block-q4k":
        shapes, nseq, L = LLAMA3_8B, args.calib_seqs or 128, args.seq_len or 2048
    else:
        shapes, nseq, L = TINY, args.calib_seqs or 32, args.seq_len or 512
    nseq_local = nseq // world  # contiguous shard, remainder dropped (quant.py:177-179)
    params = sum(R * C for R, C, _ in shapes.values())
    costs = {n: float(R) * C * (C + 128) for n, (R, C, _) in shapes.items()}
    split_names = dist_utils.row_split_names(costs, world)
    owners = dist_utils.assign_owners({n: c for n, c in costs.items() if n not in split_names}, world)
    owners.update({n: ROW_SPLIT for n in split_names})

    W16 = make_weights(shapes, dev)
    X = make_inputs(shapes, nseq_local, L, dev, seed=1 + rank)
    # four SYRK launches per grid (32 sequences = 65536 tokens each): the kernel sustains more over short token
    # ranges (same box, TFLOP/s of the SYRK in this bench: 128 seq/launch 1236-1252, 64: 1267, 43: 1278, 32: 1280-1291,
    # 21-24: 1281-1297, 16: 1266-1280 -- below 32 the extra read-modify-write of H and the launches eat the gain)
    hb = args.hessian_batch or min(32, nseq_local)
    hws = torch.empty(sum(ops.workspace_bytes(_cabi.WS_H_ACCUMULATE, 0, x.shape[-1], hb * L) for x in X.values()),
                      dtype=torch.uint8, device=dev)
    streams = [torch.cuda.Stream(dev) for _ in range(args.streams)] if args.streams > 0 else None
    torch.cuda.synchronize()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        quantize_block(shapes, W16, X, owners, rank, world, hbatch=hb, hws=hws, streams=streams, row_chunks=args.row_chunks)
    sync()
    # dominant kernel (the fp16 MFMA SYRK of the Hessian accumulation) timed live with HIP
    if world > 1 and os.environ.get("GQ_BENCH_VERIFY") == "1":  # every rank must hold the same results
        outv = quantize_block(shapes, W16, X, owners, rank, world, hbatch=hb, hws=hws, streams=streams,
                              row_chunks=args.row_chunks)
        sync()
        for name in sorted(outv):
            cs_ = outv[name].double().sum().reshape(1)
            lo_, hi_ = cs_.clone(), cs_.clone()
            dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
            assert float(lo_) == float(hi_) and float(outv[name].float().abs().sum()) > 0, f"{name}: ranks disagree"
        if rank == 0:
            print("verify: all ranks hold identical results", file=sys.stderr)
    # events on its launch stream, inside the timed region
    _cabi.prof_enable(["syrk"])
    keep = {}
    t0 = time.perf_counter()
    for i in range(args.steps):
        quantize_block(shapes, W16, X, owners, rank, world, keep=keep if i == args.steps - 1 else None, hbatch=hb,
                       hws=hws, streams=streams, row_chunks=args.row_chunks)
    sync()
    dt = time.perf_counter() - t0
    prof = _cabi.prof_collect(busy=True)
    _cabi.prof_enable([])
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if args.breakdown and rank == 0:
        _cabi.prof_enable(None)
        quantize_block(shapes, W16, X, owners, rank, world, hbatch=hb, hws=hws, streams=streams, row_chunks=args.row_chunks)
        torch.cuda.synchronize()
        bd = _cabi.prof_collect(busy=True)  # GQ_PROF_DUMP=<file> also writes the interval timeline
        _cabi.prof_enable([])
        tot = sum(v[0] for v in bd.values())
        for k, (ms, n, busy) in sorted(bd.items(), key=lambda kv: -kv[1][0]):
            print(f"  {k:22s} {ms:10.2f} ms  {n:6d} launches  {100 * ms / tot:5.1f} %  busy {busy:8.2f} ms", file=sys.stderr)

    if rank == 0:
        # roofline of the dominant kernel: algorithmic MFMA flops of the upper-triangular SYRK at 128x128
        # granularity = 2 * T * 128*128 * ntiles over the live HIP-event time.  The two SYRK grids
        # of a step overlap on two streams, so the divisor is the UNION of their launch intervals
        # (busy_ms); sum_launch_ms / launches is the plain per-launch average rocprof reports.
        syrk_sum_ms, syrk_n, syrk_ms = prof.get("syrk", (0.0, 0, 0.0))
        flops = 0.0
        for inp, x in X.items():
            C = x.shape[-1]
            nt = C // 128
            flops += x.shape[0] * args.steps * 2.0 * L * 128 * 128 * (nt * (nt + 1) // 2)  # all launches together
        ach = flops / (syrk_ms * 1e-3) / 1e12 if syrk_ms > 0 else None
        roof = {"bound": "mfma", "kernel": "syrk16_256n_kernel<f16> (gq_h_accumulate_grouped)",
                "achieved": round(ach, 2) if ach else None, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_F16_MFMA_TFLOPS, 4) if ach else None,
                # PMC cannot be read live: L2-miss reads per SYRK launch (rocprofv3 --pmc FETCH_SIZE on this very
                # command, x2 gfx950 correction; profiles/r01_syrk_pmc.txt), average of the step's launches
                "traffic": SYRK_TRAFFIC_GB_PER_LAUNCH.get(hb) if (args.workload == "llama3-8b-block-q4k" and world == 1
                                                               and not args.calib_seqs and not args.seq_len) else None,
                "traffic_unit": "GB/launch",
                "launches": syrk_n, "busy_ms_per_step": round(syrk_ms / args.steps, 3),
                "avg_launch_ms": round(syrk_sum_ms / max(syrk_n, 1), 4),
                "share_of_step": round(syrk_ms / 1e3 / dt, 3)}
        line = {
            "metric": "Mparams/s GPTQ-quantized", "value": round(params * args.steps / dt / 1e6, 2),
            "unit": "Mparams/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: 7 Linears of one block ({params / 1e6:.1f} M params), "
                                   f"{nseq}x{L}-token calibration ({hb} sequences per Hessian launch), block_size 128, "
                                   f"rel_damp 0.01, nstep 20",
                       "calib_seqs_per_rank": nseq_local, "parallelism": f"calib-dp{world}+matrix-fanout",
                       "owners": {n: (f"rows/{world}" if o == ROW_SPLIT else o) for n, o in owners.items()} if world > 1 else "rank0"},
            "wall_s_llama3_8b_32_blocks_extrapolated": round(dt / args.steps * 32, 2)
            if args.workload.startswith("llama3") else None,
            "roofline": roof,
            "trailing_update_roofline": trailing_update_roofline(shapes, W16),
            "cpu_baseline": None if args.no_cpu_baseline else cpu_baseline(shapes, W16, keep),
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
