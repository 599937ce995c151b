"""Block schedule: how the Linears of ONE transformer block travel the hot path on an MI355X.

The reference walks the handles of a block one after the other (quantizer.py:248-275): seven Hessians, seven
factorisations, seven column loops, each a chain of small dependent launches on one stream.  This module is the
product's replacement for that loop -- `Quantizer._quant_group` and `bench.py` both call it, nothing else
schedules kernels:

  hook side    `feed(name, x)` is the body of the forward hook (quantizer.py:226-232).  Linears fed by the SAME
               tensor (q/k/v, gate/up) are detected and share one Hessian.  Activations are buffered by the
               handles; `sample_done()` (after every calibration sample) folds the buffers of ALL distinct
               inputs into their Hessians with grouped SYRK launches once they hold `flush_tokens` tokens:
               the narrow inputs in one grid, then the widest input alone on the chip.  `pre_hook(name)` is the
               same feed from a forward PRE-hook; it ends forward #1 at the last hooked Linear (r04).
  quantize()   phase 0  the remaining tokens are folded in;
                        [N>1] ONE collective per distinct Hessian, widest first (gptq.py:131-132 does one all-reduce
                        per handle): a reduce to the owner when all its Linears belong to one rank, else an all-reduce;
               phase 1  every input group is an independent chain (h_prepare -> per Linear: working copy,
                        column loop, dequantize) and runs on its own HIP stream, widest chain first: the
                        single-workgroup leaves of one factorisation and the 64-CU column-loop kernels overlap
                        with the GEMMs of the other chains.  A follower reuses its leader's U when both weights
                        have the same all-zero columns (U depends on H, the dead channels and that set only,
                        gptq.py:307-313); the sets are compared on the device when the block's first sample has
                        shown who shares what, and the answer reaches the host long before quantize();
               phase 2  [N>1] ONE all-gather per block: every rank contributes what it computed (the matrices it
                        owns, its row slices of the row-split ones) as one packed byte buffer; the dequantized
                        weight is written back (quantizer.py:257-264).
quantize() never synchronises the host with the device (dense Linears): everything is ordered by streams and
events, so the caller's next launches (the block's second forward, or the next block of a benchmark) queue up
behind the chains.  The device-side flags of the reused factorisations are kept in `BlockSchedule.unverified`,
sent to pinned host memory behind an event after every block and asserted by `verify()` as they arrive (the
Quantizer: after every block without waiting, and once more, waiting, at the end of the model).
"""
import contextlib
import os
import sys
import time
from typing import Any, Callable, Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import dist_utils
from . import ops as _ops
from .gptq import GPTQ
from .model_utils import ForwardInterrupt
from .quant_utils import GGML_QUANT_SIZES
from .quant_utils import GGMLQuantizationType, dequantize_linear_weight

_stream_pool: Dict[Any, List["torch.cuda.Stream"]] = {}


def _chain_streams(device, n: int):
    """`n` persistent side streams of `device` (none for CPU tensors: the CPU tests run the chains in line)."""
    if n <= 0 or torch.device(device).type != "cuda":
        return []
    pool = _stream_pool.setdefault(torch.device(device), [])
    while len(pool) < n:
        # (measured and dropped: a high-priority stream for the costliest chain -- no gain; r03: side lanes confined to
        # 128-224 CUs by hipExtStreamCreateWithCUMask so that the costliest chain always finds free CUs -- its
        # factorisation takes 25 ms inside a step against 14.5 alone -- 115-121 ms per step against 95: kernels of
        # CU-masked queues start far slower on this platform, as the far-update helper found in r02)
        pool.append(torch.cuda.Stream(device))
    return pool[:n]


class _Lane:
    """One chain's execution lane: a HIP stream with event fork/join against the caller's stream, or nothing."""

    def __init__(self, stream, main):
        self.stream, self.main = stream, main

    def wait(self, event):
        if self.stream is not None and event is not None:
            self.stream.wait_event(event)

    def run(self):
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def join(self, tensors=()):
        """The caller's stream waits for this lane; tensors born here stay valid for the caller's stream."""
        if self.stream is None:
            return
        ev = torch.cuda.Event()
        ev.record(self.stream)
        self.main.wait_event(ev)
        for t in tensors:
            if t is not None and t.is_cuda:
                t.record_stream(self.main)


def _zero_columns(w: torch.Tensor) -> torch.Tensor:
    """bool[C]: columns of the Linear's weight that are zero in every row (gptq.py:307-308 looks for them)."""
    return (w.detach().reshape(w.shape[0], -1) == 0).all(dim=0)


class BlockSchedule:
    unverified: List[torch.Tensor] = []  # device flags of speculative U reuses, see verify()
    # (Measured and removed, r04: starting the other chains only when the costliest chain's factorisation is through -- it
    # takes 25 ms next to three other chains' GEMMs and 14.6 ms alone -- 96.4 vs 88.9 ms per step: what the others have to do
    # does NOT fit under its column loop.)

    def __init__(self, layers: Dict[str, nn.Module], make_handle: Callable[[nn.Module, str], GPTQ],
                 n_streams: Optional[int] = None, verbose: bool = False):
        self.handles: Dict[str, GPTQ] = {n: make_handle(l, n) for n, l in layers.items()}
        self._zero_cols = {n: _zero_columns(l.weight) for n, l in layers.items()}
        self._sharing = None  # ([follower names], host bool tensor "zero-column sets differ", event)
        for h in self.handles.values():
            h._scheduled = True  # the handle leaves threshold flushes to sample_done()
        self._seen: Dict[Any, Any] = {}  # per block call: input identity -> (leader handle, tensor kept alive)
        self._fired: List[str] = []      # pre_hook(): the handles fed in the current sample, in order
        self._last_fired: Optional[str] = None
        self._other_after = False        # an UN-hooked Linear of the block ran after the last hooked one (other_pre_hook)
        self._n_samples = 0
        self.n_streams = int(os.environ.get("GQ_CHAIN_STREAMS", 4)) if n_streams is None else n_streams
        self.stack = os.environ.get("GQ_STACK", "1") != "0"  # Linears that share U walk the columns together
        self.verbose = verbose
        self.stats = {"syrk_launches": 0, "allreduce_bytes": 0, "reused_U": 0, "own_U": 0, "refactorised": 0}
        self.owners: Dict[str, Any] = {}  # name -> owner rank or "rows/<world>" of the last quantize()

    # ------------------------------------------------------------------ hook side
    def hook(self, name: str):
        def _hook(_, inp, out):
            self.feed(name, inp[0])
        return _hook

    def pre_hook(self, name: str, interrupt: bool = True):
        """The same feed as hook(), from a forward PRE-hook -- the input of a Linear exists before its GEMM runs -- so that
        forward #1 can stop at the last hooked Linear: the reference discards that forward's output (quantizer.py:150-151),
        and for a Llama block the last Linear is down_proj, 27 % of the layer's GEMM flops (+ the residual add).  Which
        Linear is last is LEARNED: the block's first sample runs to its end; if every handle fired exactly once there, the
        later samples raise ForwardInterrupt right after feeding the Linear that fired last -- provided every other handle
        has fired in that sample too (MoE experts fire data-dependently: then the forward just continues) AND no Linear
        outside the hooked set ran after it in that first sample (other_pre_hook: with a --quantizable_modules regex that
        leaves out the block's structurally last Linear, forward #1 runs in full, as in the reference).
        What a caller can see of the interrupt: modules after the last Linear (for a Llama block: the residual add) do not
        run in forward #1, whose output the reference discards; Quantizer additionally skips forward #2 of the LAST block, so
        the data_loader's tensors hold the input of the last block instead of its output afterwards (nothing in the reference
        reads them: quantizer.py:181-198 quantizes the post-block modules from their weights).  `--full_forward1` runs both."""
        def _hook(_, inp):
            self.feed(name, inp[0])
            self._fired.append(name)
            self._other_after = False
            if interrupt and name == self._last_fired and len(self._fired) == len(self.handles) \
                    and len(set(self._fired)) == len(self._fired):
                self.stats["forward1_interrupts"] = self.stats.get("forward1_interrupts", 0) + 1
                raise ForwardInterrupt
        return _hook

    def other_pre_hook(self):
        """Forward pre-hook for the block's Linears that are NOT quantized (not matched by --quantizable_modules): the first
        sample records whether one of them runs after the last hooked Linear; if so forward #1 is never interrupted."""
        def _hook(_, inp):
            if self._n_samples == 0:
                self._other_after = True
        return _hook

    def feed(self, name: str, x: torch.Tensor) -> None:
        """Body of the reference's forward hook (quantizer.py:229): handles[name].update(inp[0]), plus the
        detection of Linears that see the very same tensor."""
        h = self.handles[name]
        key = (x.data_ptr(), tuple(x.shape), tuple(x.stride()), x.dtype, x._version)
        hit = self._seen.get(key)
        if hit is not None and hit[0].d_col == h.d_col and hit[0] is not h:
            leader = hit[0]
            assert h.shared_H_with in (None, leader), "input sharing pattern changed between samples"
            assert h.H is None, "handle switched from own Hessian to a shared one"
            h.shared_H_with = leader
            leader._has_followers = True
        else:
            assert h.shared_H_with is None, "input sharing pattern changed between samples"
            self._seen[key] = (h, x)
        h.update(x)

    def sample_done(self) -> None:
        """Called after every calibration sample (one forward of the block)."""
        self._seen.clear()
        if self._n_samples == 0 and len(self._fired) == len(self.handles) and len(set(self._fired)) == len(self._fired) \
                and not self._other_after:
            self._last_fired = self._fired[-1]  # every Linear fired once: later samples may stop there (pre_hook)
        self._n_samples += 1
        self._fired = []
        if self._sharing is None:
            self._publish_sharing()
        # only the inputs that hold a fold's worth of tokens are folded: an MoE expert sees top_k / experts of the tokens and
        # reaches 64 Ki four times later than the attention inputs of its block -- folding it along with them made its SYRK
        # launches four times as many and a quarter as long (tile prologue / epilogue and the read-modify-write of H per flop)
        due = [h for h in self.leaders() if h._fill >= h.flush_tokens]
        if due:
            self.flush(only=due)

    def _publish_sharing(self) -> None:
        """After the block's first sample: for every follower, "its weight's all-zero columns differ from its
        leader's" is computed on the device and copied to pinned host memory behind an event."""
        name_of = {id(h): n for n, h in self.handles.items()}
        followers = [n for n, h in self.handles.items() if h.shared_H_with is not None]
        host = ev = None
        if followers:
            neq = torch.stack([(self._zero_cols[n] != self._zero_cols[name_of[id(self.handles[n].shared_H_with)]]).any()
                               for n in followers])
            if neq.is_cuda:
                host = torch.empty(len(followers), dtype=torch.bool, pin_memory=True)
                host.copy_(neq, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(neq.device))
            else:
                host = neq
        self._sharing = (followers, host, ev)

    def _needs_own_factorisation(self) -> Dict[str, bool]:
        if self._sharing is None or not self._sharing[0]:
            return {}
        followers, host, ev = self._sharing
        if ev is not None:
            ev.synchronize()  # recorded during the first sample of the block: long complete
        return dict(zip(followers, host.tolist()))

    _staged_checks: List[Any] = []  # (pinned host bool, event): flags on their way to the host

    @classmethod
    def stage_verify(cls) -> None:
        """Send the flags collected so far to pinned host memory behind an event (no synchronisation)."""
        flags, cls.unverified = cls.unverified, []
        if not flags:
            return
        anyf = torch.stack([f.reshape(()) for f in flags]).any().reshape(1)
        if anyf.is_cuda:
            host = torch.empty(1, dtype=torch.bool, pin_memory=True)
            host.copy_(anyf, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(anyf.device))
            cls._staged_checks.append((host, ev))
        else:
            cls._staged_checks.append((anyf, None))

    @classmethod
    def verify(cls, wait: bool = True) -> None:
        """No reused factorisation saw a different column set.  wait=False (the Quantizer, once per block): only the
        flags that have already arrived are looked at -- a block's flags are checked while the next block runs, so a
        mismatch stops the run one block later at most, without ever stalling the host; wait=True (end of the
        model, tests): everything, one host read."""
        cls.stage_verify()
        keep = []
        for host, ev in cls._staged_checks:
            if ev is not None and not wait and not ev.query():
                keep.append((host, ev))
                continue
            if ev is not None:
                ev.synchronize()
            if bool(host.item()):
                cls._staged_checks = []
                raise RuntimeError("a Linear reused its leader's Cholesky factor although their dead / zero-column sets "
                                   "differ (gq_w_prepare flag)")
        cls._staged_checks = keep

    @classmethod
    def discard_checks(cls) -> None:
        """Drop every pending flag (Quantizer.quantize's finally: a run that raised must not leave flags behind for
        the next Quantizer of the process)."""
        cls.unverified = []
        cls._staged_checks = []

    def leaders(self) -> List[GPTQ]:
        return [h for h in self.handles.values() if h.shared_H_with is None]

    def _fold(self, grids: List[List[GPTQ]]) -> None:
        """One grouped SYRK launch per grid (on the current stream)."""
        for grp in grids:
            args = [h._flush_args() for h in grp]
            _ops.h_accumulate_grouped([a[0] for a in args], [a[1] for a in args], [a[2] for a in args],
                                      [a[3] for a in args])
            self.stats["syrk_launches"] += 1
            for h in grp:
                h._flush_done()

    def flush(self, only: Optional[List[GPTQ]] = None) -> None:
        """Fold every leader's buffered activations into its Hessian: grouped SYRK launches (<= 8 problems per
        grid, one activation dtype per grid): up to eight inputs in one grid; more than eight (MoE blocks): the narrow
        inputs in grids of eight first and the widest input alone.
        (Measured and removed, r02 / r04: postponing the narrow inputs' folds to quantize(), next to the widest chain --
        0.7-1 % per step for 6 GB of kept activations)"""
        todo = [h for h in self.leaders() if h._fill > 0 and (only is None or any(h is o for o in only))]
        if not todo:
            return
        todo.sort(key=lambda h: (-h.d_col, id(h)))
        grids: List[List[GPTQ]] = []
        if 1 < len(todo) <= 8 and len({(h._pending_dtype(), h._pending_device()) for h in todo}) == 1:
            # r04: up to eight inputs of one dtype share ONE grid, widest first: the K-split of the last round then levels the
            # tail of all of them at once (a dense 8B block: 2004 tiles = 7.83 rounds, executed as 7.83; as two grids the three
            # 4096-wide inputs alone ran 1.667 rounds for 1.594 of work): SYRK 53.7-54.2 against 54.2-54.5 ms per step
            rest, grids_tail = todo, []
        elif len(todo) > 1 and todo[0].d_col > todo[1].d_col:
            rest, grids_tail = todo[1:], [[todo[0]]]
        else:
            rest, grids_tail = todo, []
        by_kind: Dict[Any, List[GPTQ]] = {}
        for h in rest:
            by_kind.setdefault((h._pending_dtype(), h._pending_device()), []).append(h)
        for grp in by_kind.values():
            grids += [grp[i:i + 8] for i in range(0, len(grp), 8)]
        grids += grids_tail
        self._fold(grids)

    # ------------------------------------------------------------------ multi-rank agreement
    def _agree_on_sharing(self) -> None:
        """MoE experts receive a data-dependent set of tokens per rank; a rank that routed nothing to an expert
        never fires its hooks and cannot know that w1 and w3 share their input.  The sharing pattern decides how
        many collectives a rank issues, so it is made rank-invariant before any of them: every rank contributes
        the leader index it observed (-1: never fired) and adopts the maximum."""
        names = list(self.handles)
        idx = {id(self.handles[n]): i for i, n in enumerate(names)}
        dev = next(iter(self.handles.values())).W_device
        v = torch.full((len(names),), -1, dtype=torch.int64, device=dev)
        for i, n in enumerate(names):
            h = self.handles[n]
            if h.shared_H_with is not None:
                v[i] = idx[id(h.shared_H_with)]
            elif h.H is not None or h._fill > 0:
                v[i] = i
        mine = v.clone()
        dist_utils.collective_calls["small_all_reduce"] += 1
        dist_utils.collective_bytes["small_all_reduce"] += v.numel() * 8
        torch.distributed.all_reduce(v, op=torch.distributed.ReduceOp.MAX)
        got, mine = v.tolist(), mine.tolist()
        for i, n in enumerate(names):
            h = self.handles[n]
            assert mine[i] in (-1, got[i]), f"{n}: ranks observed different input sharing ({mine[i]} vs {got[i]})"
            if mine[i] == -1 and got[i] not in (-1, i):
                leader = self.handles[names[got[i]]]
                h.shared_H_with = leader
                leader._has_followers = True

    # ------------------------------------------------------------------ quantize
    @torch.no_grad()
    def quantize(self, qtypes: Dict[str, GGMLQuantizationType], writeback: bool = True,
                 extra: Optional[Callable[[str, GPTQ, tuple], Any]] = None) -> Dict[str, Tuple[torch.Tensor, ...]]:
        """-> {name: (qweight, super_group_scale, group_scale_quant, super_group_zero, group_zero_quant)} on every
        rank, in the reference's order (gptq.py:295); with `writeback` the dequantized matrix replaces
        layer.weight.data (quantizer.py:257-264).  `extra(name, handle, result)` runs on the chain's stream right
        after a Linear's column loop on the rank that computed it (bench.py packs the GGUF bytes there).

        (Measured and removed, r04: a deferred join -- the caller's stream runs the attention half of forward #2 behind the
        attention chains while the MLP chains are still going, DESIGN.md 6b -- the down_proj chain, ~1500 small dependent
        launches, takes 3x as long next to the forward's GEMMs as alone: quant_group + forward #2 5.35 s against 5.2 s.)"""
        handles = self.handles
        world, rank = dist_utils.get_world_size(), dist_utils.get_rank()
        for n, h in handles.items():
            if qtypes[n] == GGMLQuantizationType.Q3_K:
                h.act_order = False  # reference gptq.py:204-206
        if world > 1:
            costs = {n: float(h.d_row) * h.d_col * (h.d_col + 128) for n, h in handles.items()}
            # a matrix that outweighs a fair share is quantized by every rank on its own rows (U replicated);
            # the rest are handed out whole
            split = {n for n in dist_utils.row_split_names(costs, world) if not handles[n].act_order}
            for n in split:
                handles[n].row_split = True
            for n, r in dist_utils.assign_owners({n: c for n, c in costs.items() if n not in split}, world).items():
                handles[n].owner_rank = r
            if any(h.allow_no_samples for h in handles.values()):
                self._agree_on_sharing()
            self.owners = {n: (f"rows/{world}" if h._row_split_active() else h.owner_rank) for n, h in handles.items()}
            # reduce-to-owner (SURVEY section 5 iii): a Hessian whose Linears -- the leader and its followers -- all belong to
            # ONE rank is needed there only; row-split matrices are factorised by every rank: all-reduce
            need: Dict[int, set] = {}
            for n, h in handles.items():
                lead = h.shared_H_with or h
                need.setdefault(id(lead), set()).update(range(world) if h._row_split_active() else (h.owner_rank,))
            for h in self.leaders():
                ranks = need.get(id(h), set())
                h.reduce_to = next(iter(ranks)) if len(ranks) == 1 else None
        dev = next(iter(handles.values())).W_device
        on_gpu = torch.device(dev).type == "cuda"
        main = torch.cuda.current_stream(dev) if on_gpu else None

        # ---- phase 0: the rest of the tokens, then one all-reduce per distinct Hessian, widest first
        ready: Dict[int, Any] = {}
        self.flush()
        for h in sorted(self.leaders(), key=lambda h: -h.d_col):
            h.sync_hessian()
            if world > 1:
                self.stats["allreduce_bytes"] += dist_utils.hessian_payload_bytes(h.d_col)
                if on_gpu:
                    ready[id(h)] = torch.cuda.Event()
                    ready[id(h)].record(main)
        for h in handles.values():
            h.sync_hessian()  # followers adopt their leader's H (no collective)
        start = None
        if on_gpu and world == 1:
            start = torch.cuda.Event()
            start.record(main)

        # ---- phase 1: one chain per distinct Hessian, each on its own stream, the costliest first
        chains: Dict[int, List[str]] = {}
        for n, h in handles.items():
            if h.owner_rank == rank or h._row_split_active():
                chains.setdefault(id(h.shared_H_with or h), []).append(n)

        def chain_cost(ns):
            return (not any(handles[n]._row_split_active() for n in ns),
                    -sum(float(handles[n].d_row) * handles[n].d_col ** 2 for n in ns))

        own = self._needs_own_factorisation()
        order = sorted(chains.values(), key=chain_cost)
        # the costliest chain stays on the caller's stream (it ends last anyway), the others get side streams
        on_main = 1
        # a chain whose column loop keeps its far updates on the library's helper stream brings a hardware queue
        # of its own: one lane fewer here (five queues cost more than the overlap gains).  Only when
        # every chain still gets a lane of its own or nearly so (a dense block: 4 chains); a Mixtral block's 20 chains
        # need all the lanes, and the helper stays off
        lent = 1 if on_gpu and 2 < self.n_streams and len(order) <= self.n_streams and any(
            _ops.uses_helper_stream(h.d_row, h.d_col, h.block_size) for h in handles.values()
            if h.owner_rank == rank and not h._row_split_active()) else 0
        helper_was = _ops.far_helper_enable(bool(lent)) if on_gpu else None
        try:
            streams = [None] * on_main + _chain_streams(dev, min(self.n_streams - lent, len(order)) - on_main)
            results: Dict[str, tuple] = {}
            deq: Dict[str, torch.Tensor] = {}
            lanes = []
            trace = "sched" in os.environ.get("GQ_TRACE", "")
            t_host = time.perf_counter()
            # lanes: longest-processing-time-first over all lanes; the costliest chain comes first and lands on lane 0 = the
            # caller's stream (a dense block: four chains over three lanes + the library's helper; a Mixtral block: 20 chains over 4 lanes)
            cost = [sum(float(handles[n].d_row) * handles[n].d_col ** 2 + float(handles[n].d_col) ** 3 / 3 for n in names)
                    for names in order]
            lane_of, load = [0] * len(order), [0.0] * max(len(streams), 1)
            for k in range(len(order)):
                lane_of[k] = min(range(len(streams)), key=lambda i: (load[i], i)) if streams else 0
                load[lane_of[k]] += cost[k]
            enq = list(range(len(order)))
            if len(streams) >= 4 and len(order) > 2 * len(streams):
                # many chains per lane (a Mixtral block: eight 14336-wide w2 chains -- long strings of small dependent
                # launches -- and eight stacked w1 / w3 chains -- short and GEMM-heavy): the upper half of the lanes walk
                # their chains cheapest first, so that at any time half the lanes are in latency-bound chains and half in
                # GEMM-bound ones instead of all in the same kind (measured, same box: 306.7 against 311.3 ms per block; a lane of its own
                # for the GEMM-heavy chains and three for the others: 318.7)
                per = [[k for k in range(len(order)) if lane_of[k] == i] for i in range(len(streams))]
                for i in range(len(streams) // 2, len(streams)):
                    per[i].reverse()
                enq = [per[i][j] for j in range(max(len(x) for x in per)) for i in range(len(streams)) if j < len(per[i])]
            for k in enq:
                names = order[k]
                if trace:
                    print(f"  [sched] +{1e3 * (time.perf_counter() - t_host):7.2f} ms: enqueue chain {k} {names} on lane {lane_of[k]}",
                          file=sys.stderr)
                lane = _Lane(streams[lane_of[k]] if streams else None, main)
                lead = handles[names[0]].shared_H_with or handles[names[0]]
                lane.wait(ready.get(id(lead), start))
                born = []

                def computed(n, h, res):
                    if n in own and h._pending_mismatch is not None:
                        BlockSchedule.unverified.append(h._pending_mismatch)
                        h._pending_mismatch = None
                        self.stats["reused_U"] += 1
                    elif own.get(n, False):
                        self.stats["refactorised"] += 1
                    results[n] = res
                    born.extend(res)
                    # what the row-split fallback (recompute_whole, on the caller's stream after the exchange) reads was born
                    # on this lane too: U, the column flags, the re-search counter and the working copy
                    born.extend(t for t in (h._last_U, h._last_cf, getattr(h, "_researches", None), h.W) if torch.is_tensor(t))
                    if world == 1 and writeback:
                        deq[n] = dequantize_linear_weight(qtypes[n], *res, out_dtype=h.layer.weight.data.dtype)
                        born.append(deq[n])
                    if extra is not None:
                        out = extra(n, h, res)
                        born.extend(t for t in (out if isinstance(out, (tuple, list)) else (out,)) if torch.is_tensor(t))

                with lane.run():
                    # the leader first: it factorises, the followers reuse its U.  Linears that are KNOWN to share the
                    # factorisation (same input, same dead / zero-column sets) and the same grid walk the columns
                    # together, stacked by rows (GPTQ.compute_stacked): q / k / v and gate / up cost one walk each
                    for grp in self._stack_groups(sorted(names, key=lambda n: handles[n].shared_H_with is not None), qtypes, own):
                        if self.verbose:
                            for n in grp:
                                print(f"[rank {rank}] Quantizing {n} with {qtypes[n].name}.")
                        if len(grp) > 1:
                            hs = [handles[n] for n in grp]
                            self.stats["stacked"] = self.stats.get("stacked", 0) + len(grp)
                            for n, h, res in zip(grp, hs, GPTQ.compute_stacked(hs, qtypes[grp[0]])):
                                computed(n, h, res)
                            continue
                        n = grp[0]
                        h = handles[n]
                        h.make_working_copy()
                        # follower with the same zero columns as its leader: reuse (flag kept for verify());
                        # different: own factorisation; unknown (first fed after the first sample): checked below
                        computed(n, h, h.compute(qtypes[n], defer_check=True, own_U=own.get(n, False)))
                lanes.append((lane, born))
            for lane, born in lanes:
                lane.join(born)
        finally:
            if helper_was is not None:
                _ops.far_helper_enable(helper_was)
        if trace:
            print(f"  [sched] +{1e3 * (time.perf_counter() - t_host):7.2f} ms: all chains enqueued", file=sys.stderr)

        # followers whose sharing was not known after the block's first sample (an expert that got its first token
        # later): the device flag is read now -- the one host sync, MoE blocks only
        pending = [(n, handles[n]._pending_mismatch) for n in results if handles[n]._pending_mismatch is not None]
        if pending:
            flags = torch.cat([f.reshape(1) for _, f in pending]).tolist()
            for (n, _), bad in zip(pending, flags):
                h = handles[n]
                h._pending_mismatch = None
                if bad:
                    self.stats["refactorised"] += 1
                    h.make_working_copy()
                    results[n] = h.compute(qtypes[n], own_U=True)
                    deq.pop(n, None)
                else:
                    self.stats["reused_U"] += 1
        self.stats["own_U"] = len(results) - self.stats["reused_U"]

        # ---- phase 2: exchange, write-back.  ONE all-gather per block: every rank packs what it computed (whole
        # matrices it owns, its row slices of the row-split ones) into one byte buffer whose layout every rank can
        # derive from the owner map alone (the reference broadcasts five tensors per Linear from rank 0,
        # gptq.py:287-293: 35 collectives per Llama block, 140 for a Mixtral block).
        if world > 1:
            out = self._exchange_block(handles, qtypes, results, world, rank)
        else:
            out = {n: results[n] for n in handles}
        for n, h in handles.items():
            if writeback:
                w = deq.get(n)
                if w is None:
                    w = dequantize_linear_weight(qtypes[n], *out[n], out_dtype=h.layer.weight.data.dtype)
                h.layer.weight.data = w
            h.reset()
        return out

    def _stack_groups(self, names: List[str], qtypes, own: Dict[str, bool]) -> List[List[str]]:
        """The Linears of one chain (one input tensor; the leader first), cut into the groups that walk the columns
        together: the leader and the followers whose dead / zero-column sets are known to equal the leader's, per grid
        (GPTQ.stack_key), at most eight to a group.  Everything else -- followers with sets of their own, followers whose
        sharing became known too late, act_order or row-split matrices -- walks alone, in the old order.
        GQ_STACK=0 (or BlockSchedule.stack = False): nobody is stacked."""
        if not self.stack:
            return [[n] for n in names]
        groups: List[List[str]] = []
        open_: Dict[Any, List[str]] = {}
        for n in names:
            h = self.handles[n]
            key = h.stack_key(qtypes[n])
            known = (h.shared_H_with is None and h._has_followers) or (n in own and not own[n])
            if key is None or not known:
                groups.append([n])
                continue
            g = open_.get(key)
            if g is None or len(g) >= 8:
                g = open_[key] = []
                groups.append(g)
            g.append(n)
        return groups

    @staticmethod
    def _piece_layout(q_type, rows: int, cols: int):
        """[(dtype, shape, byte offset)] of the five result tensors of `rows` rows, each padded to 256 bytes; total."""
        bits, clamp, scale_maxq, group_size, supergroup_size, sz_dtype, q_dtype = GGML_QUANT_SIZES[q_type]
        shapes = ((q_dtype, (rows, cols)), (torch.float16, (rows, cols // supergroup_size)),
                  (sz_dtype, (rows, cols // group_size)), (torch.float16, (rows, cols // supergroup_size)),
                  (sz_dtype, (rows, cols // group_size)))
        lay, off = [], 0
        for dt, shp in shapes:
            lay.append((dt, shp, off))
            n = shp[0] * shp[1] * torch.empty(0, dtype=dt).element_size()
            off += (n + 255) & ~255
        return lay, off

    def _exchange_block(self, handles, qtypes, results, world: int, rank: int):
        per_rank: List[List[Tuple[str, int, int]]] = [[] for _ in range(world)]
        for n, h in handles.items():  # same order, same owner map on every rank
            if h._row_split_active():
                for r in range(world):
                    r0, r1, _ = dist_utils.row_slice(h.d_row, r, world)
                    if r1 > r0:
                        per_rank[r].append((n, r0, r1))
            else:
                per_rank[h.owner_rank].append((n, 0, h.d_row))
        layouts = [[(n, r0, r1) + self._piece_layout(qtypes[n], r1 - r0, handles[n].d_col) for n, r0, r1 in lst]
                   for lst in per_rank]
        sizes = [sum(x[4] for x in lst) for lst in layouts]
        # the row-split matrices' re-search counts (GPTQ.compute: quant_utils.py:250-252 evaluated per row slice) ride in the
        # same buffer: one int32 per split matrix at the end of every rank's chunk -- no collective of their own
        split = [n for n, h in handles.items() if h._row_split_active()]
        tail = (4 * len(split) + 255) & ~255 if split else 0
        body = max(max(sizes), 256)
        nbytes = body + tail
        dev = next(iter(handles.values())).W_device
        buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        off = 0
        for n, r0, r1, lay, tot in layouts[rank]:
            for t, (dt, shp, o) in zip(results[n], lay):
                k = shp[0] * shp[1] * t.element_size()
                buf[off + o:off + o + k].view(dt).view(shp).copy_(t)
            off += tot
        if split:
            buf[body:body + 4 * len(split)].view(torch.int32).copy_(torch.cat([handles[n]._researches for n in split]))
        gathered = dist_utils.all_gather_bytes(buf, nbytes)
        self.stats["allgather_bytes"] = self.stats.get("allgather_bytes", 0) + nbytes
        redo = []
        if split:
            # the one host read of the exchange (4 ranks and up only): which split matrices must be quantized whole
            counts = gathered[:, body:body + 4 * len(split)].view(torch.int32).reshape(world, len(split)).sum(dim=0).tolist()
            redo = [n for n, c in zip(split, counts) if c != 0]
        parts: Dict[str, List[List[torch.Tensor]]] = {n: [] for n in handles}
        for r in range(world):
            off = 0
            for n, r0, r1, lay, tot in layouts[r]:
                views = []
                for dt, shp, o in lay:
                    k = shp[0] * shp[1] * torch.empty(0, dtype=dt).element_size()
                    views.append(gathered[r, off + o:off + o + k].view(dt).view(shp))
                parts[n].append(views)
                off += tot
        out: Dict[str, tuple] = {}
        for n in handles:
            ps = parts[n]
            if n in redo:
                # some rank's slice decided the panel-wide `continue` on its own rows: every rank quantizes the whole matrix
                # (same H, same U on every rank -> identical results, the N = 1 bytes; no second exchange)
                self.stats["row_split_redone"] = self.stats.get("row_split_redone", 0) + 1
                handles[n].row_split_redone = True
                out[n] = handles[n].recompute_whole(qtypes[n])
                continue
            # a whole matrix: views into the gathered buffer (no copy); row slices: concatenated in rank order
            out[n] = tuple(ps[0]) if len(ps) == 1 else tuple(torch.cat([p[i] for p in ps], dim=0) for i in range(5))
        return out
