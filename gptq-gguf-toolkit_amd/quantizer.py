"""Model-walk driver: drop-in for the reference's quant/gptq/src/quantizer.py `Quantizer`.

Same constructor, same walk (embed -> blocks in order -> lm_head), same hook protocol
(forward hooks feed inp[0] of every regex-matched Linear to its handle), same data.pth
schema (quantizer.py:268-275).  MI355X-first differences, none of which changes results:
  * the Linears of a block go through `block_schedule.BlockSchedule`: Linears fed by the SAME input tensor
    (q/k/v, gate/up) share one Hessian accumulation and one all-reduce instead of 3/2 identical ones, the
    Hessians of a block are accumulated by grouped SYRK launches, and the independent input groups quantize
    concurrently on their own HIP streams;
  * with world_size > 1 the Linears of a block are assigned to owner ranks (LPT on
    R*C*(C+128)) and quantize concurrently; the 5 result tensors are broadcast from the
    owner (the reference computes everything on rank 0, gptq.py:158);
  * every data.pth is written ONCE, by the rank its index maps to (the reference lets every rank write the same
    file), by a writer thread behind a copy stream, so that the device-to-host copies and the file writes of
    block i overlap with the forwards of block i+1; `quantize()` returns after the last file of every rank is
    closed (final barrier).
`Quantizer.timing` holds the split of the last `quantize()` call (host seconds per phase and, with
GQ_TIMING=gpu, HIP-event seconds per phase on the main stream).
"""
import atexit
import os
import queue
import sys
import threading
import time
from typing import Any, Dict, Iterable, List, Optional

import torch
import torch.nn as nn

from . import dist_utils
from . import ops as _ops
from .block_schedule import BlockSchedule
from .forward_fused import fused_forward, level_of, reset_verdicts
from .gptq import GPTQ
from .model_utils import ForwardInterrupt, InputCollector, LINEAR_LAYERS, _to, select_layers
from .quant_utils import GGML_QUANT_SIZES, GGMLQuantizationType, dequantize_linear_weight


def _same(a, b) -> bool:
    if torch.is_tensor(a) or torch.is_tensor(b):
        return torch.is_tensor(a) and torch.is_tensor(b) and a.shape == b.shape and a.dtype == b.dtype and \
            (a.data_ptr() == b.data_ptr() or bool(torch.equal(a, b)))
    if isinstance(a, (tuple, list)) and isinstance(b, (tuple, list)):
        return len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    if isinstance(a, dict) and isinstance(b, dict):
        return a.keys() == b.keys() and all(_same(a[k], b[k]) for k in a)
    return a is b or a == b


def _batch_block_inputs(input_args, input_kwargs, batch: int):
    """Merge runs of up to `batch` consecutive captured block inputs into one call: the hidden states (first
    positional argument, or kwargs["hidden_states"]) are concatenated along dim 0; everything else must be IDENTICAL
    across the run (position ids / embeddings, masks, flags) and is taken from its first sample.  Samples that do
    not fit (another length, another mask) stay calls of their own."""
    def hidden(a, kw):
        return a[0] if len(a) > 0 else kw.get("hidden_states")

    def rest(a, kw):
        return (tuple(a[1:]), {k: v for k, v in kw.items() if k != "hidden_states" or len(a) > 0})

    out_a, out_kw, i, n = [], [], 0, len(input_args)
    while i < n:
        j = i + 1
        h0 = hidden(input_args[i], input_kwargs[i])
        while j < n and j - i < batch and torch.is_tensor(h0) and h0.dim() >= 2:
            hj = hidden(input_args[j], input_kwargs[j])
            if not (torch.is_tensor(hj) and hj.shape == h0.shape and hj.dtype == h0.dtype
                    and _same(rest(input_args[i], input_kwargs[i]), rest(input_args[j], input_kwargs[j]))):
                break
            j += 1
        if j - i == 1:
            out_a.append(input_args[i])
            out_kw.append(input_kwargs[i])
        else:
            cat = torch.cat([hidden(input_args[k], input_kwargs[k]) for k in range(i, j)], dim=0)
            if len(input_args[i]) > 0:
                out_a.append((cat,) + tuple(input_args[i][1:]))
                out_kw.append(input_kwargs[i])
            else:
                out_a.append(input_args[i])
                out_kw.append(dict(input_kwargs[i], hidden_states=cat))
        i = j
    return out_a, out_kw


def _own_storage(t):
    """torch.save writes a tensor's WHOLE storage: a view into a larger buffer (a staging slot, the gathered block
    buffer of the N > 1 exchange) is copied out first."""
    return t if t.untyped_storage().nbytes() == t.numel() * t.element_size() else t.clone()


def _write_data_pth(save_dir, name, q_type, qweight, d, s, dmin, m):
    os.makedirs(os.path.join(save_dir, name), exist_ok=True)
    qweight, d, s, dmin, m = (_own_storage(t) for t in (qweight, d, s, dmin, m))
    torch.save({"q_type": int(q_type), "qweight": qweight, "super_group_scale": d, "super_group_zero": dmin,
                "group_scale_quant": s, "group_zero_quant": m}, os.path.join(save_dir, name, "data.pth"))


def _layout_of(tensors, off):
    """(offset, bytes, dtype, shape) of each tensor packed from `off` on, 256-byte aligned; and the end offset."""
    layout = []
    for t in tensors:
        n = t.numel() * t.element_size()
        layout.append((off, n, t.dtype, tuple(t.shape)))
        off += (n + 255) & ~255
    return layout, off


def _slot_views(slot, layout):
    return [slot[off:off + n].view(dt).view(shape) for off, n, dt, shape in layout]


def _writer_process(save_dir, slots, inbox, freeq, outbox):
    """Body of the writer PROCESS.  `slots` are the parent's pinned staging buffers (shared memory): a message names
    a slot and the layout of a module's five tensors inside it; they are copied out, the slot goes back to the parent,
    then torch.save runs here -- with this process's interpreter lock, not the parent's."""
    failed = None
    busy = 0.0
    try:
        torch.set_num_threads(1)
        while True:
            item = inbox.get()
            if item is None:
                break
            t0 = time.perf_counter()
            try:
                if item[0] == "slots":  # one staging slot holding the tensors of one or more modules
                    _, sid, modules = item
                    name = modules[0][0]
                    try:
                        hosts = [[v.clone() for v in _slot_views(slots[sid], layout)] for _, _, layout in modules] \
                            if failed is None else []
                    finally:
                        freeq.put(sid)  # ALWAYS handed back: the parent must never wait on a writer that failed
                    for (name, q_type, _), host in zip(modules, hosts):
                        if failed is None:
                            _write_data_pth(save_dir, name, q_type, *host)
                    hosts = None
                    host = None
                else:  # a module larger than a slot: its host tensors came through the queue
                    _, name, q_type, host = item
                    if failed is None:
                        _write_data_pth(save_dir, name, q_type, *host)
            except BaseException as e:  # disk full, unwritable save_dir, ...: reported at once, later items are dropped
                if failed is None:
                    failed = f"{name}: {e!r}"
                    outbox.put(("error", failed))
            del item
            host = None
            busy += time.perf_counter() - t0
        outbox.put(("ok", busy) if failed is None else ("error", failed))
    except BaseException as e:  # reported to the parent
        outbox.put(("error", repr(e)))


class _Saver:
    """data.pth writer (quantizer.py:267-275) off the critical path.  The calling thread copies a block's result
    tensors into a pinned staging slot with copy kernels on its own stream (put_many: why not a side thread's stream);
    the slots live in shared memory; a copier thread waits for the copies on the host and a separate PROCESS copies the
    slot out and runs torch.save.  Why a process: torch.save holds the interpreter lock for much of its 0.7 s per GB and
    Llama-3-8B leaves 8.7 GB of data.pth behind -- written from a thread of this process, the launches of the next
    block's forwards stall; why pinned slots: fresh pageable host tensors cost 6-9 s in page faults.
    `_Saver.USE_PROCESS = False` keeps everything in-process; `sync=True` (CPU tensors) writes in line like the reference."""

    USE_PROCESS, KERNEL_COPY, INLINE = True, True, True
    _teardowns: List[threading.Thread] = []  # deferred close() tails still running (see close / _join_teardowns)

    @staticmethod
    def _join_teardowns(timeout: float = 90.0) -> None:
        while _Saver._teardowns:
            _Saver._teardowns.pop().join(timeout=timeout)

    def __init__(self, save_dir: str, sync: bool):
        self.save_dir, self.sync = save_dir, sync
        self.q: "queue.Queue" = queue.Queue()
        self.err: Optional[BaseException] = None
        self.busy_s = 0.0
        self.thread = None
        self.proc = self.inbox = self.outbox = self.freeq = None
        self.slots: List[torch.Tensor] = []
        self._writer_dead = False
        self._final = None
        # class-level switches (profiles/ probes flip them; the measured alternatives are in DESIGN.md 6b): torch.save in a
        # writer PROCESS (False: in this process's copier thread), copy kernels through the slot's device mapping (False:
        # hipMemcpyAsync), staged by the caller's thread on its own stream (False: by the copier thread on a stream of its own
        # -- which puts the run into the slow mode of DESIGN.md 6b)
        self.use_process, self._kernel_copy, self._inline = _Saver.USE_PROCESS, _Saver.KERNEL_COPY, _Saver.INLINE
        self.n_slots = max(2, int(os.environ.get("GQ_SAVE_SLOTS", 3)))
        self._ready = threading.Event()
        self._poll_lock = threading.Lock()
        self._pinned_ids = set()
        self.slot_bytes = int(os.environ.get("GQ_SAVE_SLOT_MB", 704)) << 20  # embed_tokens of Llama-3 in Q4_K: 657 MB

    def _start_process(self):
        import torch.multiprocessing as tmp
        ctx = tmp.get_context("spawn")
        _Saver._join_teardowns()  # a previous saver's slots (2.1 GB of /dev/shm) are gone before this one asks for room
        # the slots are files in /dev/shm: ftruncate succeeds on a small tmpfs (Docker's default is 64 MB) and the
        # first write into the slot then kills the process with SIGBUS -- ask before allocating
        need = self.n_slots * self.slot_bytes + (64 << 20)
        try:
            st = os.statvfs("/dev/shm")
            free = st.f_bavail * st.f_frsize
        except OSError:
            free = 0
        if free < need:
            raise OSError(f"/dev/shm has {free >> 20} MiB free, the writer process needs {need >> 20} MiB "
                          f"(GQ_SAVE_SLOT_MB={self.slot_bytes >> 20})")
        # slots: shared-memory storages created as such (share_memory_() on an ordinary tensor would copy 704 MB each);
        # nothing touches their pages until they are pinned
        def new_slot():
            try:
                return torch.empty(0, dtype=torch.uint8).set_(torch.UntypedStorage._new_shared(self.slot_bytes))
            except Exception:
                return torch.empty(self.slot_bytes, dtype=torch.uint8).share_memory_()
        self.slots = [new_slot() for _ in range(self.n_slots)]
        self._registered = []
        self._pinned_ids = set()
        self.inbox, self.outbox, self.freeq = ctx.Queue(), ctx.Queue(), ctx.Queue()
        self.proc = ctx.Process(target=_writer_process, args=(self.save_dir, self.slots, self.inbox, self.freeq, self.outbox),
                                daemon=True)
        self.proc.start()  # the child imports torch while the slots are pinned here
        for sid, t in enumerate(self.slots):  # pin the shared pages: the copy kernels write through their device mapping
            try:
                if int(torch.cuda.cudart().cudaHostRegister(t.data_ptr(), t.numel(), 0)) == 0:
                    self._registered.append(t)
                    self._pinned_ids.add(sid)
                else:
                    self._kernel_copy = False  # an unpinned slot has no device mapping: hipMemcpyAsync stages it
            except Exception:
                self._kernel_copy = False
            self.freeq.put(sid)
            self._ready.set()  # the first slot is enough to start with (embed_tokens); the others follow

    def _loop(self):
        stream = None
        while True:
            item = self.q.get()
            if item is None:
                return
            try:
                t0 = time.perf_counter()
                if item[0] == "staged":
                    # the main thread's copy kernels filled slot `sid` behind `ev` (put_many): wait for them on the host
                    # and hand the slot to the writer
                    _, sid, modules, ev = item
                    sent = False
                    try:
                        ev.synchronize()
                        self.inbox.put(("slots", sid, modules))
                        sent = True
                    finally:
                        if not sent:
                            self.freeq.put(sid)
                    self.busy_s += time.perf_counter() - t0
                    continue
                if item[0] == "host":
                    _, name, q_type, host = item
                    if self.proc is not None and not self._writer_dead:
                        self.inbox.put(("host", name, int(q_type), host))
                    else:
                        _write_data_pth(self.save_dir, name, q_type, *host)
                    self.busy_s += time.perf_counter() - t0
                    continue
                _, name, q_type, tensors, ev, dev = item
                if stream is None:
                    stream = torch.cuda.Stream(dev)
                layout, off = _layout_of(tensors, 0)
                # host-side wait (no interpreter lock held): nothing is parked on a hardware queue before the results exist
                ev.synchronize()
                with torch.cuda.stream(stream):
                    sid = self._get_slot() if (self.proc is not None and off <= self.slot_bytes) else None
                    if sid is not None:
                        sent = False
                        try:
                            for v, t in zip(_slot_views(self.slots[sid], layout), tensors):
                                if self._kernel_copy:
                                    _ops.stage_to_host(v, t.contiguous(), stream)
                                else:
                                    v.copy_(t.contiguous(), non_blocking=True)
                            stream.synchronize()
                            self.inbox.put(("slots", sid, [(name, int(q_type), layout)]))
                            sent = True
                        finally:
                            if not sent:  # an exception between taking the slot and handing it over: no leak
                                self.freeq.put(sid)
                    else:
                        host = [t.cpu() for t in tensors]
                        if self.proc is not None and not self._writer_dead:
                            self.inbox.put(("host", name, int(q_type), host))
                        else:
                            _write_data_pth(self.save_dir, name, q_type, *host)
                del tensors, item
                self.busy_s += time.perf_counter() - t0
            except BaseException as e:  # surfaced by close()
                self.err = self.err or e

    def _poll_writer(self) -> None:
        """Non-blocking look at the writer: an error message or a dead process switches to in-thread writing (the
        failed file is reported by close(); nothing ever waits on a writer that cannot answer)."""
        if self.proc is None or self._writer_dead:
            return
        with self._poll_lock:  # the main thread (put_many) and the copier thread both look
            try:
                while True:
                    status, info = self.outbox.get_nowait()
                    if status == "error":
                        self._writer_failed(info)
                    else:
                        self._final = (status, info)
            except queue.Empty:
                pass
            if self._final is None and not self._writer_dead and not self.proc.is_alive():
                self._writer_failed(f"writer process exited with code {self.proc.exitcode}")

    def _writer_failed(self, info) -> None:
        self._writer_dead = True
        self.err = self.err or RuntimeError(f"data.pth writer process failed: {info}")

    def _get_slot(self):
        """A free staging slot, or None when the writer process is gone (the caller then writes in this thread)."""
        while True:
            self._poll_writer()
            if self._writer_dead:
                return None
            try:
                return self.freeq.get(timeout=1.0)
            except queue.Empty:
                continue

    def warm_up(self, device) -> None:
        """Start the writer process and pin its slots now (0.3 s), in the background of the capture forward."""
        if self.sync or self.thread is not None or torch.device(device).type != "cuda":
            return
        def boot():
            try:
                if self.use_process:
                    self._start_process()
            except BaseException as e:  # no writer process (small /dev/shm, spawn failure): this thread writes, as
                # _Saver.USE_PROCESS = False does -- a start-up problem of an optimisation is logged, not raised after hours
                dist_utils.print_on_main(f"[gq] data.pth writer process not started ({e}); writing from a thread")
                self.proc = None
                self.slots = []
            self._ready.set()
            self._loop()

        self.thread = threading.Thread(target=boot, name="gq-data-pth-copier", daemon=True)
        self.thread.start()

    def put(self, name, q_type, tensors):
        self.put_many([(name, q_type, tensors)])

    def put_many(self, items):
        """Queue the tensors of some modules (one transformer block's Linears, or embed / lm_head alone) for writing.
        Default path: THIS thread launches copy kernels (gq_stage_to_host) on its current stream, right behind the
        kernels that produce the tensors, into one pinned staging slot for the whole group; the copier thread only waits
        for them (on the host) and passes the slot to the writer process.  Measured on Llama-3-8B: a second thread
        launching ANYTHING on a stream of its own while this thread issues the forwards -- one 16-element fill per
        module is enough -- costs 2-3 s of a 12 s run (forward #1 5.8 -> 7.5 s, the chains 1.1 -> 1.6 s); GPU-side the
        copies are 5 ms per block either way.  In that slow mode the large kernels run as fast as ever (SYRK 1.77 vs 1.68 s
        per run) while the launches that hang on cross-stream events do not: the single-workgroup Cholesky leaves take
        574 vs 305 ms, the column-loop kernels 793 vs 590 ms (bench.py, GQ_BENCH_WM_PROF) -- dependent dispatch got slower,
        not the kernels; the cause inside the runtime is not identified (hardware-queue count, copy engine, PCIe and the
        allocator are ruled out: DESIGN.md 6b).  (_Saver.INLINE = False: the copier thread copies on its own stream.)"""
        if not items:
            return
        if self.sync or not items[0][2][0].is_cuda:
            for name, q_type, tensors in items:
                t0 = time.perf_counter()
                _write_data_pth(self.save_dir, name, q_type, *[t.cpu() for t in tensors])
                self.busy_s += time.perf_counter() - t0
            return
        dev = items[0][2][0].device
        if self.thread is None:
            self.warm_up(dev)
        cur = torch.cuda.current_stream(dev)
        self._ready.wait()  # the writer process is up (or known to be absent)
        inline = (self._inline and self.proc is not None and self._kernel_copy and not self._writer_dead)
        group, size = [], 0

        def flush():
            nonlocal group, size
            if not group:
                return
            # May wait for the writer to hand a slot back (that needs only work which is already queued: the copy kernels,
            # the copier's host-side wait, the writer's clone).  Waiting is the lesser evil: sending the group down the
            # copier thread's own-stream path instead was tried -- one early miss and the rest of the run is in the slow
            # mode (13.4 vs 12.1 s).  Three slots: a Llama block needs one per 350 ms, the writer frees one in 130 ms.
            sid = self._get_slot()
            if sid is None:
                for name, q_type, tensors in group:
                    self._put_one(name, q_type, tensors, cur, dev)
            else:
                sent = False
                try:
                    modules, off = [], 0
                    for name, q_type, tensors in group:
                        layout, off = _layout_of(tensors, off)
                        modules.append((name, int(q_type), layout))
                        for v, t in zip(_slot_views(self.slots[sid], layout), tensors):
                            if sid in self._pinned_ids:
                                _ops.stage_to_host(v, t.contiguous(), cur)
                            else:  # a slot that could not be pinned has no device mapping
                                v.copy_(t.contiguous(), non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(cur)
                    self.q.put(("staged", sid, modules, ev))
                    sent = True
                finally:
                    if not sent:
                        self.freeq.put(sid)
            group, size = [], 0

        for name, q_type, tensors in items:
            n = _layout_of(tensors, 0)[1]
            if inline and n > self.slot_bytes:
                # larger than a slot (a 70B model's embed_tokens): copied out by THIS thread, synchronously -- twice per
                # model -- rather than by the copier on a stream of its own
                flush()
                self.q.put(("host", name, q_type, [t.cpu() for t in tensors]))
                continue
            if not inline:
                flush()
                self._put_one(name, q_type, tensors, cur, dev)
                continue
            if size + n > self.slot_bytes:
                flush()
            group.append((name, q_type, tensors))
            size += n
        flush()

    def _put_one(self, name, q_type, tensors, cur, dev):
        ev = torch.cuda.Event()
        ev.record(cur)
        self.q.put(("copy", name, q_type, tensors, ev, dev))

    def close(self):
        t0 = time.perf_counter()
        self.close_split = {}
        if self.thread is not None:
            self.q.put(None)
            while self.thread.is_alive():  # the copier never blocks for good (_get_slot polls the writer), but look anyway
                self.thread.join(timeout=5.0)
                self._poll_writer()
            self.thread = None
        self.close_split["copier_drained"] = round(time.perf_counter() - t0, 4)
        if self.proc is not None:
            self._poll_writer()
            status, info = "error", "no answer"
            try:
                self.inbox.put(None)
                deadline = time.time() + 600
                while self._final is None and time.time() < deadline:
                    try:
                        st, inf = self.outbox.get(timeout=0.05)
                        if st == "error" and not self._writer_dead:
                            self._writer_failed(inf)
                        self._final = (st, inf)
                    except queue.Empty:
                        if not self.proc.is_alive():
                            break
                if self._final is not None:
                    status, info = self._final
                elif not self.proc.is_alive():
                    info = f"writer process exited with code {self.proc.exitcode}"
            except Exception as e:
                info = f"writer process did not answer: {e!r}"
            self.close_split["writer_answered"] = round(time.perf_counter() - t0, 4)
            # every file is on disk once the writer has answered.  What is left -- the child's interpreter exit (it imported
            # torch: 0.2-0.4 s) and un-pinning three 704 MB slots -- needs nobody's attention: a daemon thread does it, the
            # slots stay referenced by it until then
            proc, registered, slots = self.proc, list(getattr(self, "_registered", [])), self.slots

            def _teardown():
                try:
                    proc.join(timeout=60)
                finally:
                    for t in registered:
                        try:
                            torch.cuda.cudart().cudaHostUnregister(t.data_ptr())
                        except Exception:
                            pass
                    slots.clear()
            if status == "ok" and not self._writer_dead:
                # joined by the next _Saver before it sizes /dev/shm, and at interpreter exit BEFORE the HIP runtime is
                # finalised (atexit hooks run ahead of module teardown): the thread never calls into a dying runtime
                th = threading.Thread(target=_teardown, daemon=True, name="gq-saver-teardown")
                _Saver._teardowns.append(th)
                th.start()
            else:
                _teardown()
            self._registered = []
            self.proc = self.inbox = self.outbox = self.freeq = None
            self.slots = []
            self.close_split["returned"] = round(time.perf_counter() - t0, 4)
            if status == "ok" and not self._writer_dead:
                self.writer_busy_s = float(info)
            else:
                self.err = self.err or RuntimeError(f"data.pth writer process failed: {info}")
        if self.err is not None:
            raise self.err


atexit.register(_Saver._join_teardowns, 30.0)


class _Phases:
    """Where the time of quantize() goes: host seconds per phase, plus GPU seconds per phase from HIP events on
    the main stream when GQ_TIMING=gpu (the events are read once, at the end)."""

    def __init__(self, device):
        self.host: Dict[str, float] = {}
        self.gpu_events: List[Any] = []
        self.use_gpu = os.environ.get("GQ_TIMING") == "gpu" and torch.device(device).type == "cuda"
        self.device = device
        self._t = time.perf_counter()
        self._e = self._event()

    def _event(self):
        if not self.use_gpu:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream(self.device))
        return e

    def mark(self, phase: str):
        now = time.perf_counter()
        self.host[phase] = self.host.get(phase, 0.0) + now - self._t
        self._t = now
        if self.use_gpu:
            e = self._event()
            self.gpu_events.append((phase, self._e, e))
            self._e = e

    def result(self) -> Dict[str, Any]:
        out = {"host_s": {k: round(v, 4) for k, v in self.host.items()}}
        if self.use_gpu:
            torch.cuda.synchronize(self.device)
            g: Dict[str, float] = {}
            for phase, a, b in self.gpu_events:
                g[phase] = g.get(phase, 0.0) + a.elapsed_time(b) * 1e-3
            out["gpu_s"] = {k: round(v, 4) for k, v in g.items()}
        return out


def _first(x):
    return x[0] if isinstance(x, (tuple, list)) else x


class Quantizer:
    def __init__(self, model: nn.Module, data_loader: Iterable, quantizable_modules: str,
                 quantizer_kwargs: Dict[str, Any], pre_block_modules: List[str], post_block_modules: List[str],
                 block_modules: str, save_dir: str, quant_non_block_modules: bool = False,
                 device: Optional[torch.device] = None, cpu_offload_modules: bool = False,
                 cpu_offload_activations: bool = False, verbose: bool = False, non_block_fp32: bool = False,
                 calibration_batch: int = 1, fused_forward="exact", interrupt_forward1: bool = True) -> None:
        self.model = model
        self.data_loader = data_loader
        self.quantizable_modules = quantizable_modules
        self.quantizer_kwargs = quantizer_kwargs
        self.pre_block_modules = pre_block_modules
        self.post_block_modules = post_block_modules
        self.block_modules = block_modules
        self.device = device
        self.cpu_offload_modules = cpu_offload_modules
        self.cpu_offload_activations = cpu_offload_activations
        self.quant_non_block_modules = quant_non_block_modules
        self.verbose = verbose
        self.save_dir = save_dir
        self.non_block_fp32 = non_block_fp32
        # beyond the reference (which runs one calibration sample per block forward, quantizer.py:150-151): samples
        # per block forward.  Same Hessians in exact arithmetic (GPTQ.update weighs a batch by its size,
        # gptq.py:86-112); one Llama-3-8B layer forward takes 0.91 instead of 1.16 ms per sequence at 4.
        self.calibration_batch = max(1, int(calibration_batch))
        # forward #1 of a block (quantizer.py:150-151) only feeds the Hessian hooks, its output is discarded: stop it at the
        # last hooked Linear (learned on the block's first sample; BlockSchedule.pre_hook).  Same Hessians, same bytes saved.
        # The same switch drops forward #2 of the LAST block, whose outputs nothing reads.
        self.interrupt_forward1 = bool(interrupt_forward1)
        # beyond the reference (which runs the HF eager modules, quantizer.py:293): HIP kernels for the elementwise
        # modules of the block forward (forward_fused.py).  "exact" (default): rotary embedding, SwiGLU and RMSNorm, each
        # bit-identical to HF eager (RMSNorm verified at run time on this PyTorch before it is trusted) -- the saved
        # tensors do not change; "all" / True: RMSNorm's free-order kernel (<= 2 ulp) where the exact one is not
        # verified; "off" / False: none.  GQ_FUSED_FORWARD=off|exact|all overrides (A/B runs).
        self.fused_forward = level_of(os.environ.get("GQ_FUSED_FORWARD", fused_forward))

    # ------------------------------------------------------------------ walk
    @torch.no_grad()
    def quantize(self, quant_config: Dict[str, GGMLQuantizationType]) -> None:
        device = self.device or next(self.model.parameters()).device
        self._save_index = -1
        self._saver = _Saver(self.save_dir, sync=False)
        if os.environ.get("GQ_SAVE_SKIP") != "1":
            self._saver.warm_up(device)
        self._saved_names: List[str] = []
        try:
            reset_verdicts()  # every run re-verifies the forward kernels on its own first inputs
            with fused_forward(self.fused_forward if torch.device(device).type == "cuda" else "off") as patched:
                self._fused_modules = patched
                self._quantize(quant_config, device)
        finally:
            BlockSchedule.discard_checks()  # nothing of this run may leak into the next Quantizer of the process
            t0 = time.perf_counter()
            self._saver.close()
            dist_utils.barrier()  # every rank's files are on disk
            self._check_saved_files()
            if getattr(self, "_phases", None) is not None:
                self._phases.host["save_tail"] = time.perf_counter() - t0
                self.timing = self._phases.result()
                if torch.device(device).type == "cuda":
                    ms = torch.cuda.memory_stats(device)
                    self.timing["allocator"] = {"device_allocs": ms["num_device_alloc"], "segments": ms["segment.all.current"],
                                                "reserved_GiB": round(ms["reserved_bytes.all.peak"] / 2**30, 1),
                                                "active_peak_GiB": round(ms["active_bytes.all.peak"] / 2**30, 1)}
                self.timing["saver_copy_thread_busy_s"] = round(self._saver.busy_s, 4)
                self.timing["saver_close_split_s"] = getattr(self._saver, "close_split", None)
                self.timing["saver_writer_process_busy_s"] = round(getattr(self._saver, "writer_busy_s", 0.0), 4)

    def _quantize(self, quant_config: Dict[str, GGMLQuantizationType], device) -> None:
        ph = self._phases = _Phases(device)
        blocks = self.model.get_submodule(self.block_modules)
        pre_blocks = [(n, self.model.get_submodule(n)) for n in self.pre_block_modules]
        post_blocks = [(n, self.model.get_submodule(n)) for n in self.post_block_modules]
        blocks[0] = blocks[0].to(device)
        for _, m in pre_blocks:
            m.to(device)
        use_cache = getattr(self.model.config, "use_cache", None)
        if use_cache is not None:
            self.model.config.use_cache = False
        # capture the inputs of block 0 for every calibration sample (quantizer.py:78-89)
        blocks[0] = InputCollector(blocks[0], cpu_offload=self.cpu_offload_activations)
        for inp_args, inp_kwargs in self.data_loader:
            try:
                self.model(*_to(inp_args, device=device), **_to(inp_kwargs, device=device))
            except ForwardInterrupt:
                pass
        input_args, input_kwargs = blocks[0].input_args, blocks[0].input_kwargs
        blocks[0] = blocks[0].module
        if self.calibration_batch > 1:
            input_args, input_kwargs = _batch_block_inputs(input_args, input_kwargs, self.calibration_batch)
        dist_utils.barrier()
        ph.mark("capture")

        if self.quant_non_block_modules:
            for name, module in pre_blocks:
                self._quant_and_save_non_block(name, module.to(device), quant_config)
        if self.cpu_offload_modules:
            for _, m in pre_blocks:
                m.cpu()
        ph.mark("rtn_pre")
        # Where the post-block modules (lm_head) are quantized.  Their RTN reads nothing the block loop writes -- same tensors
        # and files wherever it runs (a tied lm_head still sees embed_tokens' RTN first) -- but lm_head's data.pth (657 MB for
        # Llama-3-8B: 0.6 s of clone + torch.save in the writer) is the tail of the run when it comes last, as in the
        # reference (quantizer.py:181-198).  Default: right before the LAST block, so that the file is written under that
        # block's work (save tail 0.65 -> 0.3 s).  GQ_POST_BLOCKS=last: the reference's place; =early: before the block loop
        # (same gain; with two staging slots 2 of 5 such runs fell into the slow mode described at _Saver.put_many).
        post_where = os.environ.get("GQ_POST_BLOCKS", "before_last_block") if self.quant_non_block_modules else "none"
        if post_where not in ("none", "early", "before_last_block", "last"):
            raise ValueError(f"GQ_POST_BLOCKS={post_where!r}: expected early / before_last_block / last")
        post_early = post_where == "early"
        if post_early:
            for name, module in post_blocks:
                self._quant_and_save_non_block(name, module.to(device), quant_config)
            ph.mark("rtn_post")

        for block_id, block in enumerate(blocks):
            if post_where == "before_last_block" and block_id == len(blocks) - 1:
                for name, module in post_blocks:
                    self._quant_and_save_non_block(name, module.to(device), quant_config)
                ph.mark("rtn_post")
            if self.verbose:
                dist_utils.print_on_main(f"Processing {self.block_modules} {block_id}/{len(blocks)}.")
            block = block.to(device)
            prefix = f"{self.block_modules}.{block_id}."
            layers = select_layers(self.model, prefix, self.quantizable_modules, LINEAR_LAYERS)
            handles, hooks = self._prepare_hooks_and_handles(layers, block)
            sched = self._schedule
            for a, kw in zip(input_args, input_kwargs):  # forward #1: Hessians (quantizer.py:150-151)
                try:
                    block(*_to(a, device=device), **_to(kw, device=device))
                except ForwardInterrupt:  # every hooked Linear has been fed; the reference discards the output anyway
                    pass
                sched.sample_done()  # grouped SYRK launches once the buffers hold flush_tokens tokens
            for h in hooks.values():
                h.remove()
            ph.mark("forward1+H")
            self._quant_group(handles, quant_config)
            ph.mark("quant_group")
            # forward #2: propagate (quantizer.py:161-172).  After the LAST block nothing reads the propagated activations (the
            # post-block modules are quantized from their weights alone, quantizer.py:181-198): that forward is not run
            dead = self.interrupt_forward1 and block_id == len(blocks) - 1
            for a, kw in (() if dead else zip(input_args, input_kwargs)):
                out = _first(block(*_to(a, device=device), **_to(kw, device=device)))
                if self.cpu_offload_activations:
                    out = out.cpu()
                if len(a) > 0:
                    a[0].data = out
                elif "hidden_states" in kw:
                    kw["hidden_states"] = out
                else:
                    raise ValueError("Unsupported block input format.")
            if self.cpu_offload_modules:
                block.cpu()
            del handles, hooks, sched
            self._schedule = None
            # reused-factorisation flags: this block's go to the host behind an event, earlier blocks' are read if
            # they have arrived -- a mismatch (the host-side pre-check makes it unlikely) stops the run at once
            # instead of after the whole model, and the host never waits for the device here
            BlockSchedule.verify(wait=False)
            ph.mark("forward2")
            if "blocks" in os.environ.get("GQ_TRACE", ""):  # measurement knob: per-block wall time (synchronises)
                torch.cuda.synchronize()
                now = time.perf_counter()
                ms = torch.cuda.memory_stats(device)
                print(f"[gq] block {block_id}: {1e3 * (now - getattr(self, '_t_block', now)):.1f} ms "
                      f"(saver queue {self._saver.q.qsize()}; reserved {ms['reserved_bytes.all.current'] >> 20} MiB in "
                      f"{ms['segment.all.current']} segments, {ms['num_device_alloc']} device allocs, "
                      f"active {ms['active_bytes.all.current'] >> 20} MiB)", file=sys.stderr)
                self._t_block = now

        if post_where == "last" or (post_where == "before_last_block" and len(blocks) == 0):
            for name, module in post_blocks:
                self._quant_and_save_non_block(name, module.to(device), quant_config)
        if self.quant_non_block_modules:  # wherever they ran, they ran: the packer would silently write a missing one as f16
            missing = [n for n, _ in pre_blocks + post_blocks if n not in self._saved_names]
            assert not missing, f"non-block modules never quantized: {missing}"
        if use_cache is not None:
            self.model.config.use_cache = use_cache
        BlockSchedule.verify()  # whatever is still on its way
        ph.mark("rtn_post")
        dist_utils.barrier()

    # --------------------------------------------------------- hooks / handles
    def _create_handle(self, layer, name: str = "") -> GPTQ:
        # MoE expert Linears ("...experts.<e>.w1"): the router may send them no calibration token at all, and a
        # data-dependent number per rank -- the handle then falls back to H = I and reduces H sample-weighted
        # (the reference asserts, gptq.py:126, and averages unweighted)
        return GPTQ(layer, allow_no_samples=".experts." in f".{name}.", **self.quantizer_kwargs)

    def _prepare_hooks_and_handles(self, layers: Dict[str, nn.Module], block: Optional[nn.Module] = None):
        """reference quantizer.py:221-239; the hook body lives in BlockSchedule.feed.  `block`: its Linears outside `layers`
        get a flag-only pre-hook, so that forward #1 is interrupted only when the last Linear to run is a hooked one."""
        self._schedule = BlockSchedule(layers, self._create_handle, verbose=self.verbose)
        # forward PRE-hooks: same inp[0] as the reference's forward hook (quantizer.py:229), available before the Linear's
        # GEMM, so that forward #1 can stop at the last hooked Linear (BlockSchedule.pre_hook)
        hooks = {name: layer.register_forward_pre_hook(self._schedule.pre_hook(name, self.interrupt_forward1))
                 for name, layer in layers.items()}
        if block is not None and self.interrupt_forward1:
            hooked = {id(l) for l in layers.values()}
            for i, m in enumerate(block.modules()):
                if isinstance(m, tuple(LINEAR_LAYERS)) and id(m) not in hooked:
                    hooks[f"<unhooked {i}>"] = m.register_forward_pre_hook(self._schedule.other_pre_hook())
        return self._schedule.handles, hooks

    # ------------------------------------------------------------- quantize
    def _check_saved_files(self) -> None:
        """The data.pth files are dealt to the ranks; all of them must end up in ONE directory (the packer treats a
        missing module as unquantized and would silently write it as f16).  After the barrier rank 0 looks: on a
        multi-node run with a node-local save_dir this is where it shows."""
        if not dist_utils.is_main() or os.environ.get("GQ_SAVE_SKIP") == "1" or sys.exc_info()[0] is not None:
            return
        missing = [n for n in self._saved_names if not os.path.isfile(os.path.join(self.save_dir, n, "data.pth"))]
        if missing:
            raise RuntimeError(f"{len(missing)} of {len(self._saved_names)} data.pth files are not in {self.save_dir} "
                               f"(first: {missing[0]}): save_dir must be shared by all ranks")

    def _save(self, name, q_type, qweight, d, s, dmin, m, defer: bool = False):
        self._saved_names.append(name)
        # every rank holds every result after the exchange: the files are dealt round-robin to the ranks of the
        # node (the reference lets every rank write every file, quantizer.py:267-275), so that the device-to-host
        # copies and the zip/CRC work of torch.save -- ~1 GB/s per writer, 8.7 GB for Llama-3-8B -- scale with
        # the ranks instead of serialising on rank 0 next to an 8x shorter compute phase
        self._save_index = getattr(self, "_save_index", -1) + 1
        if self._save_index % dist_utils.get_world_size() != dist_utils.get_rank() or os.environ.get("GQ_SAVE_SKIP") == "1":
            return  # (GQ_SAVE_SKIP: measurement knob -- how much of the wall time the writer costs)
        if defer:
            return (name, q_type, (qweight, d, s, dmin, m))
        self._saver.put(name, q_type, (qweight, d, s, dmin, m))

    def _quant_group(self, handles: Dict[str, GPTQ], quant_config: Dict[str, GGMLQuantizationType]):
        """reference quantizer.py:241-275: quantize every handle, write the dequantized weight back, save."""
        qtypes = {n: quant_config.get(n.split(".")[-1], GGMLQuantizationType.Q4_K) for n in handles}
        sched = self._schedule
        assert sched is not None and sched.handles is handles
        batch = []
        for n, res in sched.quantize(qtypes, writeback=True).items():
            item = self._save(n, qtypes[n], *res, defer=True)
            if item is not None:
                batch.append(item)
        self._saver.put_many(batch)  # the block's files travel in one staging slot
        self.schedule_stats = sched.stats

    def _quant_non_block_module(self, w: torch.Tensor, q_type: GGMLQuantizationType):
        """RTN for embed / lm_head (reference quantizer.py:278-330)."""
        kw = self.quantizer_kwargs
        # fp16 / bf16 weights: the reference runs make_*quants in the model dtype (quantizer.py:109,195);
        # gq_rtn_quantize reproduces that per-op rounding.  non_block_fp32 opts into an fp32 search instead.
        if w.dtype != torch.float32 and self.non_block_fp32:
            w = w.float()
        # quant_scale is forwarded like the reference does (quantizer.py:293-295; grid / maxshrink stay at
        # Quantizer.configure's defaults there, quant_utils.py:65-66).  make_k_quants ignores it.
        quant_scale = getattr(kw.get("quant_scale", "absmax"), "value", kw.get("quant_scale", "absmax"))
        if quant_scale == "mse" and w.dtype != torch.float32 and \
                q_type in (GGMLQuantizationType.Q3_K, GGMLQuantizationType.Q6_K):
            # the reference cannot run this combination: make_quants builds min_loss in fp32 (quant_utils.py:165) and
            # index_puts the model-dtype loss into it (:187) -> RuntimeError; same error class here, before any work
            raise RuntimeError(
                f"quant_scale='mse' on a {w.dtype} weight with {q_type.name}: the reference raises here (Index put "
                "requires the source and destination dtypes match, quant_utils.py:187). Use --dtype float32 or "
                "--non_block_fp32.")
        return _ops.rtn_quantize(w.contiguous(), int(q_type), kw.get("rmin", -1.0), kw.get("rdelta", 0.1),
                                 kw.get("nstep", 20), quant_scale=quant_scale)

    def _quant_and_save_non_block(self, name, module, quant_config):
        if self.verbose:
            dist_utils.print_on_main(f"Processing {name}.")
        q_type = quant_config.get(name.split(".")[-1], GGMLQuantizationType.Q6_K)  # quantizer.py:107,192
        qweight, d, s, dmin, m = self._quant_non_block_module(module.weight, q_type)
        module.weight.data = dequantize_linear_weight(q_type, qweight, d, s, dmin, m,
                                                      out_dtype=module.weight.data.dtype)
        self._save(name, q_type, qweight, d, s, dmin, m)
