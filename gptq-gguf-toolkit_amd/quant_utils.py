"""Host mirror of the reference's quant/gptq/src/quant_utils.py surface.

Same names and argument meaning (GGMLQuantizationType, GGML_QUANT_SIZES,
Quantizer.configure / get_scale_and_zero, dequantize_linear_weight), but every
numerical step is a HIP kernel behind the C ABI (ops.py); nothing is computed
with torch ops.
"""
from enum import Enum, IntEnum
from typing import Tuple

import torch

from . import ops as _ops

QK_K = 256  # gguf.constants.QK_K


class GGMLQuantizationType(IntEnum):  # reference quant_utils.py:11-16 (ggml type ids)
    Q2_K = 10
    Q3_K = 11
    Q4_K = 12
    Q5_K = 13
    Q6_K = 14


# bits, q clamp range, max scale int, group size, super-group size, dtype of scale/zero ints, dtype of qweight
# (reference quant_utils.py:19-26)
GGML_QUANT_SIZES = {
    GGMLQuantizationType.Q2_K: (2, (0, 3), 15, 16, QK_K, torch.uint8, torch.uint8),
    GGMLQuantizationType.Q3_K: (3, (-4, 3), 31, 16, QK_K, torch.int8, torch.int8),
    GGMLQuantizationType.Q4_K: (4, (0, 15), 63, 32, QK_K, torch.uint8, torch.uint8),
    GGMLQuantizationType.Q5_K: (5, (0, 31), 63, 32, QK_K, torch.uint8, torch.uint8),
    GGMLQuantizationType.Q6_K: (6, (-32, 31), 63, 16, QK_K, torch.int8, torch.int8),
}

# bytes per 256-value block (gguf.constants.GGML_QUANT_SIZES)
GGML_BLOCK_BYTES = {GGMLQuantizationType.Q2_K: 84, GGMLQuantizationType.Q3_K: 110, GGMLQuantizationType.Q4_K: 144,
                    GGMLQuantizationType.Q5_K: 176, GGMLQuantizationType.Q6_K: 210}


class QuantizationScale(str, Enum):
    ABSMAX = "absmax"
    MSE = "mse"


def _check_scale(quant_scale):
    return QuantizationScale(quant_scale)


class Quantizer:
    """Scale/min search configuration + get_scale_and_zero (reference quant_utils.py:49-145)."""

    def configure(self, bits, scale_maxq: int, group_size: int, group_type: torch.dtype, super_group_size: int,
                  quant_scale=QuantizationScale.ABSMAX, grid: int = 100, maxshrink: float = 0.80, norm: float = 2.0,
                  rmin: float = -1.0, rdelta: float = 0.1, nstep: int = 20, eps: float = 1e-9):
        self.bits = bits
        self.maxq = 2 ** bits - 1
        self.scale_maxq = scale_maxq
        self.group_size = group_size
        self.supergroup_size = super_group_size
        self.group_type = group_type
        self.rmin, self.rdelta, self.nstep, self.eps = rmin, rdelta, nstep, eps
        self.quant_scale = _check_scale(quant_scale)
        self.grid, self.maxshrink, self.norm = grid, maxshrink, norm
        if eps != 1e-9:
            raise NotImplementedError("eps other than the reference default 1e-9 is compiled into the kernels")

    def get_scale_and_zero(self, x: torch.Tensor, q_type: GGMLQuantizationType
                           ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        """x: (rows, 256) fp32 view -> (super_group_scale f16[rows], group_scale_quant[rows, 256/G],
        super_group_zero f16[rows], group_zero_quant[rows, 256/G])  -- reference return order (:145)."""
        assert x.ndim == 2 and x.shape[1] == QK_K, f"expected (rows, {QK_K})"
        if x.stride(1) != 1:
            x = x.contiguous()
        mq = dict(quant_scale=self.quant_scale.value, grid=self.grid, maxshrink=self.maxshrink)  # :164-191 (Q3_K / Q6_K)
        if self.norm != 2.0:
            raise NotImplementedError("norm other than the reference default 2.0 is compiled into the kernels")
        if x.dtype in (torch.float16, torch.bfloat16):
            # the reference runs make_*quants in the panel's dtype (quantizer.py:109,195 pass model-dtype weights):
            # gq_group_search rounds after every op the way ATen's CPU fp16 / bf16 kernels do
            _, _, d, s, dmin, m = _ops.group_search(x, int(q_type), self.rmin, self.rdelta, self.nstep, **mq)
            return d, s, dmin, m
        if x.dtype != torch.float32:
            x = x.float()
        d, s, dmin, m = _ops.scale_search(x, int(q_type), self.rmin, self.rdelta, self.nstep, **mq)
        return d, s, dmin, m


def dequantize_linear_weight(q_type, qweight, super_group_scale, group_scale_quant, super_group_zero,
                             group_zero_quant, out_dtype=torch.float32) -> torch.Tensor:
    """reference quant_utils.py:277-310; `out_dtype` folds in the caller's cast (quantizer.py:264)."""
    return _ops.dequantize(int(q_type), qweight, super_group_scale, group_scale_quant, super_group_zero,
                           group_zero_quant, out_dtype)
