"""GPTQ per-Linear handle: drop-in for the reference's quant/gptq/src/gptq.py class.

Same constructor, same `update / quantize / reset` protocol and return order
(gptq.py:29-72, :79-114, :297-302, :295); the numerical body is HIP:
  update                -> gq_h_accumulate   (K1, MFMA SYRK)
  quantization_pre_step -> all-reduce of H over RCCL (gptq.py:131-132) + fp32 working copy
  _prepare              -> gq_h_prepare      (K2 + K3)
  step                  -> gq_gptq_quantize  (K4 + K5 + K6)
"""
import os
from typing import Optional, Sequence, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn
from torch import Tensor
from torch.nn.modules.conv import _ConvNd

from . import dist_utils, model_utils
from . import ops as _ops
from .quant_utils import GGML_QUANT_SIZES, GGMLQuantizationType, QuantizationScale, _check_scale


class GPTQ:
    def __init__(self, layer: nn.Module, rel_damp: float = 1e-2, block_size: Optional[int] = None,
                 act_order: bool = False, quant_scale: str = "absmax", rmin: float = -1.0, rdelta: float = 0.1,
                 nstep: int = 20, grid: int = 100, static_groups: bool = False, verbose: bool = False,
                 allow_no_samples: bool = False):
        if act_order:
            assert static_groups  # reference gptq.py:45-46
        assert isinstance(layer, (nn.Linear, _ConvNd)), "OBC supports only linear and convolutional layers."
        self.layer = layer
        self.W = layer.weight
        self.d_row, self.d_col = model_utils.get_number_of_rows_and_cols(layer)
        self.rel_damp = rel_damp
        self.block_size = block_size or self.d_col
        self.act_order = act_order
        self.quant_scale = _check_scale(quant_scale)
        self.static_groups = static_groups
        self.grid, self.rmin, self.rdelta, self.nstep = grid, rmin, rdelta, nstep
        self.W_device, self.W_dtype, self.W_shape = self.W.device, self.W.dtype, self.W.shape
        self.H: Optional[Tensor] = None
        self.num_samples = 0
        self.verbose = verbose
        # --- beyond the reference ---
        self.owner_rank = 0            # rank that runs step(); the reference hard-codes rank 0 (gptq.py:158)
        self.row_split = False         # every rank runs step() on its own rows (dist_utils.row_split_names)
        self.row_split_redone = False  # the last row-split quantization fell back to the whole matrix (recompute_whole)
        self._researches = None        # device int32: panel-wide re-searches of this rank's row slice
        self._last_cf = None           # column flags of the last _prepare (dead / all-zero columns)
        self.reduce_to = None          # the ONE rank that needs this handle's reduced H (None: every rank -> all-reduce)
        # MoE experts may receive no calibration token at all: with allow_no_samples the handle then uses H = I
        # (round-to-nearest with the lazily computed scales) instead of the reference's assertion (gptq.py:126)
        self.allow_no_samples = allow_no_samples
        self.no_samples = False
        self.shared_H_with = None      # another handle fed by the SAME input tensor (q/k/v, gate/up)
        self._has_followers = False    # some other handle names this one in shared_H_with
        self._scheduled = False        # a BlockSchedule folds the buffers of all leaders in grouped launches
        self._reduced = False          # H already went through sync_hessian()
        self._pending_mismatch = None  # device flag of a speculative reuse of the leader's U (compute(defer_check))
        self._flag = None
        self._ws = None
        # activations are buffered (288 GB of HBM per GPU) and folded into H in long-K SYRK launches:
        # b samples at once give beta = n/(n+b), alpha = 2/(n+b), the telescoped form of b single updates
        self.flush_tokens = 1 << 16    # 65536 tokens per SYRK launch: the sweet spot measured in bench.py
        self._buf = None
        self._fill = 0                 # pending tokens (kept blocks + staged rows)
        self._staged = 0               # rows of _buf in use
        self._segs = []                # zero-copy: (tensor, version at hook time, batch size) per pending sample
        self._rag = []                 # ragged blocks kept by reference until the fold gathers them: (tensor, version)
        self._zero_copy = True         # False: copy every activation into the staging buffer at hook time
        self._buf_b = 0
        self._U_cache = None
        self._last_U = None

    # ------------------------------------------------------------------ Hessian
    @torch.no_grad()
    def update(self, input: Tensor) -> None:
        """H <- n/(n+b) H + 2/(n+b) X^T X  (reference gptq.py:79-114)."""
        batch_size = input.shape[0]
        if self.shared_H_with is not None:  # this handle's H is the leader's H (same input tensor)
            self.num_samples += batch_size
            return
        if self.H is None:
            self.H = torch.zeros((self.d_col, self.d_col), device=input.device, dtype=torch.float32)
        if isinstance(self.layer, nn.Linear):
            x = input.reshape(-1, input.shape[-1])
        else:
            unfold = nn.Unfold(self.layer.kernel_size, dilation=self.layer.dilation, padding=self.layer.padding,
                               stride=self.layer.stride)
            x = unfold(input).transpose(1, 2).flatten(0, 1)
        if x.dtype not in (torch.float16, torch.bfloat16, torch.float32):
            x = x.float()
        t = x.shape[0]
        # Zero-copy: the hook's tensor is kept (a reference, no copy) and the SYRK reads it where the forward left
        # it (gq_h_accumulate_segments) -- as long as every pending sample is one contiguous 16-bit [L, C] block of
        # the same L (a multiple of 128 tokens).  Anything else is staged into one buffer (gq_h_stage).
        keepable = self._zero_copy and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and x.is_contiguous() \
            and x.data_ptr() % 16 == 0 and t > 0
        if keepable and self._staged == 0 and not self._rag and t % 128 == 0 and self.d_col % 256 == 0 \
                and (not self._segs or (self._segs[0][0].shape == x.shape and self._segs[0][0].dtype == x.dtype)):
            self._segs.append((x, x._version, batch_size))
        elif keepable and self.d_col % 8 == 0 and self._pending_dtype() in (None, x.dtype):
            # Ragged (an MoE expert's data-dependent token count, odd sequence lengths): the tensor is kept by reference
            # too, and ONE gather launch brings every pending block into the staging buffer when the fold is due
            # (gq_h_stage_many) -- a copy per hook call is a latency-bound launch per sample and Linear
            self._rag += [(y, v) for y, v, _ in self._segs]
            self._segs = []
            self._rag.append((x, x._version))
        else:
            self._stage(x)
        self._fill += t
        self._buf_b += batch_size
        if self._fill >= self.flush_tokens and not self._scheduled:
            self.flush()

    def _gather(self) -> None:
        """The kept ragged blocks move into the staging buffer, behind the rows already there, in ONE launch."""
        if not self._rag:
            return
        for x, v in self._rag:
            if x._version != v:
                raise RuntimeError("a Linear input was modified in place after its forward hook ran; set the handle's "
                                   "_zero_copy = False to copy activations at hook time")
        xs = [x for x, _ in self._rag]
        need = self._staged + sum(x.shape[0] for x in xs)
        if self._buf is None or self._buf.dtype != xs[0].dtype or self._buf.shape[0] - self._PAD < need:
            assert self._buf is None or self._buf.dtype == xs[0].dtype or self._staged == 0
            bigger = torch.empty((need + self._PAD, self.d_col), device=xs[0].device, dtype=xs[0].dtype)
            if self._staged:
                _ops.h_stage(bigger, 0, self._buf[:self._staged])
            self._buf = bigger
        _ops.h_stage_many(self._buf, self._staged, xs)
        self._staged = need
        self._rag = []

    def _stage(self, x: Tensor) -> None:
        """Append x [t, C] to the staging buffer (behind the kept zero-copy blocks, which move into it first)."""
        if self._pending_dtype() not in (None, x.dtype):
            self.flush()  # one activation dtype per fold
        self._gather()
        pend, self._segs = [y for y, _, _ in self._segs], []
        need = self._staged + sum(y.shape[0] for y in pend) + x.shape[0]
        cap = self.flush_tokens + x.shape[0]  # a fold is due by then
        room = self._buf.shape[0] - self._PAD if self._buf is not None else 0  # rows the buffer offers (+ _PAD rows of slack)
        if self._buf is not None and self._buf.dtype == x.dtype and room < need <= cap:
            # grow geometrically (x 4: the copies add up to a third of the final size) and keep the staged rows: an MoE expert sees a data-dependent share of the tokens (and 1/N of
            # them on N ranks) -- sizing every expert's buffer for a whole fold up front cost 19 GB per rank on a Mixtral block
            bigger = torch.empty((min(cap, max(4 * room, need)) + self._PAD, self.d_col), device=x.device, dtype=x.dtype)
            if self._staged:
                _ops.h_stage(bigger, 0, self._buf[:self._staged])
            self._buf = bigger
        if self._buf is not None and (self._buf.dtype != x.dtype or need > self._buf.shape[0] - self._PAD):
            if self._staged:  # the staged rows are exactly the samples counted so far (pend is empty in this mode)
                self.flush()
                need = x.shape[0]
            self._buf = None
        if self._buf is None:
            self._buf = torch.empty((max(need, min(cap, 4 * need)) + self._PAD, self.d_col), device=x.device, dtype=x.dtype)
        for y in pend + [x]:
            _ops.h_stage(self._buf, self._staged, y)
            self._staged += y.shape[0]

    def _pending_dtype(self):
        if self._segs or self._rag:
            return (self._segs or self._rag)[0][0].dtype
        return self._buf.dtype if self._buf is not None and self._staged else None

    def _pending_device(self):
        if self._segs or self._rag:
            return (self._segs or self._rag)[0][0].device
        return self._buf.device if self._buf is not None else None

    @torch.no_grad()
    def flush(self) -> None:
        """Fold the pending activations into H (one gq_h_accumulate over all pending tokens)."""
        if self._fill == 0:
            return
        H, X, beta, alpha = self._flush_args()
        _ops.h_accumulate(H, X, beta, alpha)
        self._flush_done()

    def _flush_args(self):
        """(H, X, beta, alpha) of the pending fold; X is [T, C] or the list of kept [L, C] blocks.  b samples at
        once are the telescoped form of b single updates of gptq.py:106-112."""
        n, b = self.num_samples, self._buf_b
        self._gather()
        if self._segs:
            for x, v, _ in self._segs:
                if x._version != v:
                    raise RuntimeError("a Linear input was modified in place after its forward hook ran; set the handle's "
                                       "_zero_copy = False to copy activations at hook time")
            X = [x for x, _, _ in self._segs]
        else:
            # A ragged fold (MoE experts, odd sequence lengths) is padded with zero rows to whole turns of the SYRK's ring
            # (128 tokens): zero tokens add exact zeros to every sum, and the fold takes the kernel that reads X in place
            # (no re-layout pass, four-wave form) instead of the operand-image path
            pad = (-self._staged) % self._PAD
            if pad and self._buf.dtype in (torch.float16, torch.bfloat16) and self.d_col % 256 == 0 and self._buf.is_cuda:
                _ops.h_stage(self._buf, self._staged, self._zero_rows(pad))
            else:
                pad = 0
            X = self._buf[:self._staged + pad]
        return self.H, X, n / (n + b), 2.0 / (n + b)

    _PAD = 128  # tokens per turn of the SYRK's four-slot ring; the staging buffer keeps this many rows of slack
    _zeros: dict = {}

    def _zero_rows(self, n: int) -> Tensor:
        key = (self._buf.device, self._buf.dtype, self.d_col)
        z = GPTQ._zeros.get(key)
        if z is None:
            z = GPTQ._zeros[key] = torch.zeros((self._PAD, self.d_col), device=self._buf.device, dtype=self._buf.dtype)
        return z[:n]

    def _flush_done(self) -> None:
        self.num_samples += self._buf_b
        self._fill = 0
        self._staged = 0
        self._segs = []
        self._rag = []
        self._buf_b = 0

    def reset(self) -> None:
        """Back to the state after __init__ (reference gptq.py:116-120 frees H; every piece of state this class
        adds goes with it, so a handle can be fed and quantized again)."""
        self.W = self.layer.weight
        self.H = None
        self.num_samples = 0
        self._ws = None
        self._buf = None
        self._fill = 0
        self._staged = 0
        self._segs = []
        self._rag = []
        self._buf_b = 0
        self._reduced = False
        self.shared_H_with = None
        self._has_followers = False
        self._U_cache = None
        self._last_U = None
        self._pending_mismatch = None
        self._flag = None
        self.no_samples = False
        self.owner_rank = 0
        self.row_split = False
        self._researches = None
        self._last_cf = None
        self.reduce_to = None

    # ------------------------------------------------------------------- quantize
    @torch.no_grad()
    def quantization_pre_step(self) -> None:
        """All-reduce of H + fp32 working copy (reference gptq.py:122-143).  The dead-channel
        fix of :134-135,141 happens inside gq_h_prepare together with _prepare's masking."""
        self.sync_hessian()
        self.make_working_copy()
        self.pre_step_completed = True

    @torch.no_grad()
    def sync_hessian(self) -> None:
        """One collective per DISTINCT Hessian (reference gptq.py:131-132 does one per handle)."""
        if self.shared_H_with is not None:
            leader = self.shared_H_with
            if not leader._reduced:
                leader.sync_hessian()
            self.H = leader.H
            self._reduced = True
            return
        if self._reduced:
            return  # (a non-owner rank has dropped its copy by now: reduce_to)
        if not self.allow_no_samples:
            assert self.H is not None, "One has to process at least one sample of calibration data to run pruning"
        if self.H is None:  # this rank saw no token of this expert: it still takes part in the collective
            self.H = torch.zeros((self.d_col, self.d_col), device=self.W_device, dtype=torch.float32)
        self.flush()
        self._buf = None
        if not self._reduced:
            if self.allow_no_samples:
                # experts: sample-weighted when the ranks' token counts differ (costs one host read of the counts)
                total = dist_utils.allreduce_hessian(self.H, self.num_samples, dst=self.reduce_to)
                if total == 0:
                    self.H = torch.eye(self.d_col, device=self.W_device, dtype=torch.float32)
                    self.no_samples = True
            else:
                # every dense Linear: the reference's AVG, no host sync (reduce_to: to the one rank that needs it)
                dist_utils.allreduce_hessian(self.H, dst=self.reduce_to)
            self._reduced = True
            if self.reduce_to is not None and self.reduce_to != dist_utils.get_rank() and dist_utils.get_world_size() > 1:
                # reduced to its one owner: this rank's copy (a partial sum now) is dead weight -- 822 MB per 14336-wide
                # Hessian, eight of them per Mixtral block
                self.H = None

    @torch.no_grad()
    def make_working_copy(self, out: Optional[Tensor] = None) -> None:
        """`out`: fp32 [d_row, d_col] rows of a buffer that stacks the Linears of one input (compute_stacked)."""
        W = self.layer.weight.detach()
        if out is not None:
            assert out.shape == (self.d_row, self.d_col) and out.dtype == torch.float32 and out.is_contiguous()
            out.copy_(W.flatten(1, -1) if isinstance(self.layer, _ConvNd) else W)  # the same fp32 values as .float()
            self.W = out
            return
        W = W.clone() if W.dtype == torch.float32 else W.float()  # gptq.py:138 (clone().float(): one pass, not two)
        if isinstance(self.layer, _ConvNd):
            W = W.flatten(1, -1)
        self.W = W.contiguous()

    @torch.no_grad()
    def _prepare(self, defer_check: bool = False, own_U: bool = False) -> Tensor:
        """-> U = chol_upper(H^-1); mutates H (damping) and W (dead columns) like the reference (:304-324).
        Handles fed by the same input tensor hold the same H; U depends on (H, dead set, zero-column set
        of W) only, so a handle whose sets equal those of the handle that factorised first reuses that U
        bit-for-bit.  `defer_check`: the comparison flag stays on the device in `_pending_mismatch` (the caller
        reads it later and calls compute(own_U=True) on a mismatch) so that no host sync splits the chain."""
        leader = self.shared_H_with or self
        shared = self.shared_H_with is not None or self._has_followers
        if shared and leader._U_cache is not None and not own_U:
            U, flag, cf = leader._U_cache
            self._last_cf = cf
            mismatch = _ops.w_prepare(cf, self.W)
            if defer_check:
                self._pending_mismatch, self._flag = mismatch, flag
                return U
            if int(mismatch.item()) == 0:
                self._flag = flag
                return U
        H = self.H.clone() if shared else self.H  # every reference handle damps its own copy
        U, self._flag, cf = _ops.h_prepare(H, self.W, self.rel_damp, want_flags=True)
        self._last_cf = cf
        if shared and leader._U_cache is None and not own_U:
            leader._U_cache = (U, self._flag, cf)
        if not shared:
            self.H = H
        return U

    @property
    def issue_non_invertible(self) -> bool:
        return bool(self._flag is not None and int(self._flag.item()) != 0)

    @torch.no_grad()
    def compute(self, q_type: GGMLQuantizationType, defer_check: bool = False, own_U: bool = False):
        """Rank-local numerical body of step() (reference gptq.py:158-276); no communication."""
        if q_type == GGMLQuantizationType.Q3_K:  # reference gptq.py:204-206 mutates the handle
            self.act_order = False
            self.static_groups = False
        self.row_split_redone = False
        if self.act_order:
            return self._compute_act_order(q_type)
        U = self._prepare(defer_check, own_U)
        self._last_U = U  # for inspection (bench.py's cpu_baseline leg re-runs the column loop on the same U); dropped by reset()
        W = self.W
        if self._row_split_active():
            # every rank factorises (same reduced H => the same U, bit for bit) and walks its own rows
            r0, r1, _ = dist_utils.row_slice(self.d_row, dist_utils.get_rank(), dist_utils.get_world_size())
            # quant_utils.py:250-252 looks across ALL rows of the matrix; a slice looks across its own.  The two agree unless a
            # slice had to search a panel again (gq_gptq_quantize_slice counts those): the counts of all ranks are summed
            # after the loop (exchange / BlockSchedule._exchange_block) and a non-zero sum sends the matrix to recompute_whole
            self._researches = torch.zeros(1, dtype=torch.int32, device=self.W.device)
            if r1 <= r0:
                return tuple(t[:0] for t in self._empty_result(q_type))
            W = self.W[r0:r1]
            return _ops.gptq_quantize(W, U, int(q_type), self.block_size, self.static_groups, self.rmin, self.rdelta,
                                      self.nstep, quant_scale=self.quant_scale.value, grid=self.grid,
                                      panel_researches=self._researches)
        return _ops.gptq_quantize(W, U, int(q_type), self.block_size, self.static_groups, self.rmin,
                                  self.rdelta, self.nstep, quant_scale=self.quant_scale.value, grid=self.grid)

    @torch.no_grad()
    def recompute_whole(self, q_type: GGMLQuantizationType):
        """Row-split fallback: ALL rows of the matrix on this rank, with the factorisation compute() made -- what the owner of
        the matrix would have produced (every rank holds the same reduced H, hence the same U: the ranks' results are
        identical without an exchange).  Called on every rank when the ranks' re-search counts do not sum to zero."""
        assert self._last_U is not None and self._last_cf is not None, "recompute_whole() follows compute()"
        self.make_working_copy()
        _ops.w_prepare(self._last_cf, self.W)  # the dead / all-zero columns gq_h_prepare zeroed in the first working copy
        return _ops.gptq_quantize(self.W, self._last_U, int(q_type), self.block_size, self.static_groups, self.rmin,
                                  self.rdelta, self.nstep, quant_scale=self.quant_scale.value, grid=self.grid)

    def stack_key(self, q_type: GGMLQuantizationType):
        """Handles with equal keys that share a factorisation may be quantized in ONE walk over the columns
        (compute_stacked); None: this handle walks alone."""
        if self.act_order and q_type != GGMLQuantizationType.Q3_K:
            return None
        if self._row_split_active() or self.d_row % 64 or self.layer.weight.dtype not in (torch.float16, torch.bfloat16, torch.float32):
            return None
        return (int(q_type), self.d_col, self.block_size, bool(self.static_groups) and q_type != GGMLQuantizationType.Q3_K,
                self.rmin, self.rdelta, self.nstep, self.quant_scale, self.grid, self.rel_damp, str(self.W_device))

    @staticmethod
    @torch.no_grad()
    def compute_stacked(hs: Sequence["GPTQ"], q_type: GGMLQuantizationType):
        """compute() of several handles that share U -- the leader of an input tensor first, then followers whose dead /
        zero-column sets equal the leader's (q / k / v, gate / up, an expert's w1 / w3) -- as ONE column loop over their
        working copies stacked by rows (gq_gptq_quantize_stacked).  Rows never mix in gptq.py:222-270 and the loop is
        bound by its C dependent steps, so three Linears cost one walk; the scale search's one look across rows
        (quant_utils.py:250-252) is kept per Linear: every result equals compute()'s bit for bit.  Returns the handles'
        result tuples (row slices of the stacked outputs); follower flags stay on the device (_pending_mismatch)."""
        assert len(hs) > 1 and all(h.stack_key(q_type) == hs[0].stack_key(q_type) is not None for h in hs)
        rows = [h.d_row for h in hs]
        ends = [sum(rows[:i + 1]) for i in range(len(rows))]
        Wf = torch.empty((ends[-1], hs[0].d_col), device=hs[0].W_device, dtype=torch.float32)
        U = None
        for h, r1, n in zip(hs, ends, rows):
            if q_type == GGMLQuantizationType.Q3_K:
                h.act_order = False
                h.static_groups = False
            h.make_working_copy(out=Wf[r1 - n:r1])
            Uh = h._prepare(defer_check=True)  # the leader factorises; a follower masks its dead columns and leaves a flag
            assert U is None or Uh is U, "stacked handles must share one factorisation"
            U = Uh
            h._last_U = U
        h0 = hs[0]
        res = _ops.gptq_quantize(Wf, U, int(q_type), h0.block_size, h0.static_groups, h0.rmin, h0.rdelta, h0.nstep,
                                 quant_scale=h0.quant_scale.value, grid=h0.grid, row_ends=ends)
        for h in hs:
            # After compute() a handle's W is its dequantized fp32 working copy (the reference quantizes self.W in place).  A
            # stacked handle's copy is a row slice of Wf; keeping the VIEW would let one handle pin the whole stack until
            # reset(), aliasing the live fp16 / bf16 parameter (r05) gave readers another tensor and dtype depending on
            # whether the Linear happened to be stacked.  Contract: W is None after a stacked walk -- the
            # dequantized weights are the result tuple through dequantize_linear_weight, as BlockSchedule writes them back.
            h.W = None
        return [tuple(t[r1 - n:r1] for t in res) for r1, n in zip(ends, rows)]

    def _row_split_active(self) -> bool:
        return bool(self.row_split) and not self.act_order and dist_utils.get_world_size() > 1

    @torch.no_grad()
    def _compute_act_order(self, q_type: GGMLQuantizationType):
        """act_order=True (reference gptq.py:208-216, 233-235, 272-276; implies static_groups, :45-46).
        Columns are walked in descending diag(H) order; every column keeps the STATIC scale of its original
        group.  The reference permutes after quantization_pre_step, i.e. dead channels already count with
        H_ii = 1 and their weights are 0 (gptq.py:133-137) -- both done here before the permutation because
        gq_h_prepare, which normally applies them, only sees the permuted operands."""
        H = self.H
        diag = torch.diagonal(H).clone()
        dead = diag == 0
        diag[dead] = 1.0                                   # gptq.py:134
        self.W[:, dead] = 0                                # gptq.py:136
        # torch.argsort is not stable in the reference either; ties only occur between dead channels, whose
        # rows/columns of H and columns of W are identical, so their relative order cannot change a result
        perm = torch.argsort(diag, descending=True, stable=True)
        _, d, s, dmin, m = _ops.rtn_quantize(self.W, int(q_type), self.rmin, self.rdelta, self.nstep,
                                             quant_scale=self.quant_scale.value, grid=self.grid)  # :184-196
        Wp = self.W[:, perm].contiguous()                  # :213
        Hp = H[perm][:, perm].contiguous()                 # :214 (a copy: shared Hessians stay untouched)
        U, self._flag = _ops.h_prepare(Hp, Wp, self.rel_damp)
        qp = _ops.gptq_quantize_perm(Wp, U, int(q_type), perm.to(torch.int32).contiguous(), d, s, dmin, m,
                                     block_size=self.block_size)
        self.W = Wp                                        # the reference leaves self.W permuted
        q = qp[:, torch.argsort(perm)].contiguous()        # :274-276
        return q, d, s, dmin, m

    def _empty_result(self, q_type):
        bits, clamp, scale_maxq, group_size, supergroup_size, sz_dtype, q_dtype = GGML_QUANT_SIZES[q_type]
        dev, R, C = self.W_device, self.d_row, self.d_col
        return (torch.empty(R, C, device=dev, dtype=q_dtype),
                torch.empty(R, C // supergroup_size, device=dev, dtype=torch.float16),
                torch.empty(R, C // group_size, device=dev, dtype=sz_dtype),
                torch.empty(R, C // supergroup_size, device=dev, dtype=torch.float16),
                torch.empty(R, C // group_size, device=dev, dtype=sz_dtype))

    @torch.no_grad()
    def exchange(self, result, q_type: GGMLQuantizationType):
        """Broadcast of the 5 result tensors from the owner (reference gptq.py:287-293, src=0 there)."""
        if self._row_split_active():  # all-gather of the ranks' row slices instead of the owner's broadcast
            _, _, chunk = dist_utils.row_slice(self.d_row, dist_utils.get_rank(), dist_utils.get_world_size())
            # the slices' re-search counts ride in the first row all-gather (one extra row of the int8 / uint8 qweight chunk
            # carries the count's four bytes): no collective and no host sync of their own
            q = result[0]
            world = dist_utils.get_world_size()
            padded = torch.zeros((chunk + 1, q.shape[1]), dtype=q.dtype, device=q.device)
            padded[:q.shape[0]] = q
            padded.view(torch.uint8)[chunk, :4] = self._researches.view(torch.uint8)
            gathered = dist_utils.all_gather_rows(padded, (chunk + 1) * world, chunk + 1)
            per = gathered.view(world, chunk + 1, q.shape[1])
            counts = per[:, chunk, :4].contiguous().view(torch.uint8).view(torch.int32)
            if int(counts.sum().item()) != 0:  # some slice decided :250-252 on its own rows: the whole matrix, on every rank
                self.row_split_redone = True   # (every rank then pays the N = 1 time for this matrix: surfaced in the stats)
                return self.recompute_whole(q_type)
            qfull = per[:, :chunk].reshape(world * chunk, q.shape[1])[:self.d_row]
            return (qfull,) + tuple(dist_utils.all_gather_rows(t, self.d_row, chunk) for t in result[1:])
        if result is None:
            result = self._empty_result(q_type)
        if dist_utils.is_dist_available_and_initialized() and dist_utils.get_world_size() > 1:
            for t in result:
                dist_utils.collective_calls["broadcast"] += 1
                dist.broadcast(t, src=self.owner_rank)
        return result

    @torch.no_grad()
    def step(self, q_type: GGMLQuantizationType) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
        """-> (qweight, super_group_scale, group_scale_quant, super_group_zero, group_zero_quant)
        on every rank (reference return order, gptq.py:295)."""
        if q_type == GGMLQuantizationType.Q3_K:  # before the row-split test: compute() would reset it
            self.act_order = False
        mine = dist_utils.get_rank() == self.owner_rank or self._row_split_active()
        res = self.compute(q_type) if mine else None
        return self.exchange(res, q_type)

    def quantize(self, q_type: GGMLQuantizationType):
        self.quantization_pre_step()
        return self.step(q_type)
