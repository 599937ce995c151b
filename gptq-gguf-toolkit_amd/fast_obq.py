"""FastOBQ per-Linear handle: drop-in for EvoPress' evopress/src/fast_obq.py class (the uniform-grid GPTQ that
builds the search's layer database, evopress/src/quantizer.py:144-171).

Same constructor, same `update / quantize(bitwidth_options) / reset` protocol and the same return value
(fast_obq.py:22-57, :64-114, :214-216): ({bits: qweight u8 [R, C]}, {bits: scale [R, C/G]}, {bits: zero [R, C/G]},
perm or None).  The numerical body is the HIP path of the GPTQ handle:
  update                -> gq_h_accumulate   (MFMA SYRK; shared with gptq.GPTQ)
  quantization_pre_step -> all-reduce of H + fp32 working copy
  _prepare              -> gq_obq_h_prepare  (damping BEFORE the zero-column mask, fast_obq.py:133-141, 221-228)
  step                  -> gq_obq_quantize   once per bit width on ONE factorisation (fast_obq.py:146-200)
"""
from typing import List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn
from torch import Tensor

from . import dist_utils
from . import ops as _ops
from .gptq import GPTQ


class FastOBQ(GPTQ):
    def __init__(self, layer: nn.Module, bitwidth_options: List[int], perchannel: bool = True,
                 group_size: Optional[int] = None, sym: bool = False, rel_damp: float = 1e-2,
                 block_size: Optional[int] = None, act_order: bool = False, verbose: bool = False):
        super().__init__(layer, rel_damp=rel_damp, block_size=block_size, act_order=False, verbose=verbose)
        if not perchannel:
            # fast_obq.py:168-175 with perchannel=False hands a per-tensor grid repeated R times to a [R, 1] column:
            # never used by the reference's own caller (quantizer.py:24 fixes perchannel=True)
            raise NotImplementedError("FastOBQ: only perchannel=True (the reference caller's setting)")
        for b in bitwidth_options:
            if not 1 <= int(b) <= 8:
                raise ValueError(f"FastOBQ: bit width {b} does not fit the uint8 qweight (fast_obq.py:148)")
        self.bitwidth_options = [int(b) for b in bitwidth_options]
        self.group_size = group_size
        self.sym = sym
        self.act_order = act_order

    @torch.no_grad()
    def _prepare(self):
        """-> (w, U): fast_obq.py:133-141 + 219-232 on the device.  Unlike GPTQ there is no identity fallback: a
        Hessian that is not positive definite raises, as torch.linalg.cholesky does in the reference."""
        U, self._flag = _ops.h_prepare(self.H, self.W, self.rel_damp, obq_order=True)
        return self.W, U

    @torch.no_grad()
    def compute(self, bitwidth_options: List[int]):
        """Rank-local body of step() (fast_obq.py:146-200); no communication."""
        perm = None
        if self.act_order:  # :146-149, on the state quantization_pre_step leaves: dead diagonal 1, damped
            diag = torch.diagonal(self.H).clone()
            dead = diag == 0
            diag[dead] = 1.0
            self.W[:, dead] = 0
            diag = diag + self.rel_damp * diag.mean()
            perm = torch.argsort(diag, descending=True, stable=True)
            self.W = self.W[:, perm].contiguous()
            self.H = self.H[perm][:, perm].contiguous()
        w_all, U = self._prepare()
        self._last_U = U
        q, s, z = {}, {}, {}
        for bits in bitwidth_options:
            w = w_all.clone()
            q[bits], sc, ze = _ops.obq_quantize(w, U, bits, self.group_size or 0, self.sym, self.block_size)
            s[bits], z[bits] = sc.to(self.W_dtype), ze.to(self.W_dtype)  # :150-151 allocate them in the layer's dtype
        if int(self._flag.item()) != 0:
            raise torch.linalg.LinAlgError("FastOBQ: the damped Hessian is not positive definite")
        return q, s, z, perm

    @torch.no_grad()
    def step(self, bitwidth_options: List[int]):
        R, C, dev = self.d_row, self.d_col, self.W_device
        ng = C // (self.group_size or C)
        if dist_utils.get_rank() == self.owner_rank:
            q, s, z, perm = self.compute(bitwidth_options)
        else:
            q = {b: torch.empty(R, C, device=dev, dtype=torch.uint8) for b in bitwidth_options}
            s = {b: torch.empty(R, ng, device=dev, dtype=self.W_dtype) for b in bitwidth_options}
            z = {b: torch.empty(R, ng, device=dev, dtype=self.W_dtype) for b in bitwidth_options}
            perm = torch.empty(C, device=dev, dtype=torch.int64) if self.act_order else None
        if dist_utils.is_dist_available_and_initialized() and dist_utils.get_world_size() > 1:  # :205-210
            for b in bitwidth_options:
                for t in (q[b], s[b], z[b]):
                    dist.broadcast(t, src=self.owner_rank)
            if perm is not None:  # the reference returns a rank-local perm; every rank computes the same one there
                dist.broadcast(perm, src=self.owner_rank)
        return q, s, z, perm

    def quantize(self, bitwidth_options: Optional[List[int]] = None):
        self.quantization_pre_step()
        return self.step(list(bitwidth_options if bitwidth_options is not None else self.bitwidth_options))
