"""pack_Q2K .. pack_Q6K with the reference's signatures (packing_utils.py:33-326),
running on the GPU bit-packer kernels (gq_pack).  Differences, on purpose:
  * inputs are never modified (the reference's pack_Q3K / pack_Q6K add +4 / +32 in
    place, :94-95 / :279);
  * CPU tensors (as loaded from data.pth) are moved to the current GPU first.
Return value: C-contiguous np.uint8 [N, W/256 * type_size], as handed to
gguf_writer.add_tensor(..., raw_dtype=q_type) by pack_gptq_into_gguf.py:344-348.
"""
import numpy as np
import torch

from . import ops as _ops
from .quant_utils import GGMLQuantizationType as T


def _dev(t: torch.Tensor) -> torch.Tensor:
    if not torch.cuda.is_available():
        from ._cabi import GQError
        raise GQError("packing needs a GPU (no CPU fallback)")
    return t.contiguous().cuda()


def pack_on_device(q_type, qweights, super_group_scale, group_scale_quant, super_group_zero=None, group_zero_quant=None
                   ) -> torch.Tensor:
    """The packers' body on tensors that may already live on the GPU -> uint8 [N, W/256 * type_size] ON THE DEVICE
    (pack_gptq_into_gguf.convert keeps the five tensors of a Linear there from the upload to the packed bytes)."""
    N, W = qweights.shape
    assert W % 256 == 0, "W must be a multiple of 256"
    nsg = W // 256
    d = _dev(super_group_scale.reshape(N, nsg).to(torch.float16))
    dmin = _dev(super_group_zero.reshape(N, nsg).to(torch.float16)) if super_group_zero is not None else None
    s = _dev(group_scale_quant.reshape(N, -1))
    m = _dev(group_zero_quant.reshape(N, -1)) if group_zero_quant is not None else None
    return _ops.pack(int(q_type), _dev(qweights), d, s, dmin, m)


def _pack(q_type, qweights, super_group_scale, group_scale_quant, super_group_zero=None, group_zero_quant=None
          ) -> np.ndarray:
    return pack_on_device(q_type, qweights, super_group_scale, group_scale_quant, super_group_zero, group_zero_quant).cpu().numpy()


def pack_Q2K(qweights, super_group_scale, group_scale_quant, super_group_zero, group_zero_quant) -> np.ndarray:
    return _pack(T.Q2_K, qweights, super_group_scale, group_scale_quant, super_group_zero, group_zero_quant)


def pack_Q3K(qweights, super_group_scale, group_scale_quant) -> np.ndarray:
    return _pack(T.Q3_K, qweights, super_group_scale, group_scale_quant)


def pack_Q4K(qweights, super_group_scale, group_scale_quant, super_group_zero, group_zero_quant) -> np.ndarray:
    return _pack(T.Q4_K, qweights, super_group_scale, group_scale_quant, super_group_zero, group_zero_quant)


def pack_Q5K(qweights, super_group_scale, group_scale_quant, super_group_zero, group_zero_quant) -> np.ndarray:
    return _pack(T.Q5_K, qweights, super_group_scale, group_scale_quant, super_group_zero, group_zero_quant)


def pack_Q6K(qweights, super_group_scales, group_scale_quant) -> np.ndarray:
    return _pack(T.Q6_K, qweights, super_group_scales, group_scale_quant)


def pack_scale_min_torch(scale: torch.Tensor, zero: torch.Tensor) -> torch.Tensor:
    """12-byte 6-bit scale/min packing (packing_utils.py:8-30): bytes 4..15 of a Q4_K block."""
    assert scale.shape == zero.shape and scale.shape[1] == 8
    assert scale.dtype == torch.uint8 and zero.dtype == torch.uint8
    n = scale.shape[0]
    q = torch.zeros(n, 256, dtype=torch.uint8)
    h = torch.zeros(n, 1, dtype=torch.float16)
    blk = _pack(T.Q4_K, q, h, scale, h, zero)
    return torch.from_numpy(blk[:, 4:16].copy())


PACKERS = {T.Q2_K: pack_Q2K, T.Q3_K: pack_Q3K, T.Q4_K: pack_Q4K, T.Q5_K: pack_Q5K, T.Q6_K: pack_Q6K}


def pack_tensor(q_type, qweight, super_group_scale, group_scale_quant, super_group_zero, group_zero_quant):
    """Dispatch of pack_gptq_into_gguf.py:326-336."""
    q_type = T(int(q_type))
    if q_type in (T.Q3_K, T.Q6_K):
        return PACKERS[q_type](qweight, super_group_scale, group_scale_quant)
    return PACKERS[q_type](qweight, super_group_scale, group_scale_quant, super_group_zero, group_zero_quant)


def pack_tensor_on_device(q_type, qweight, super_group_scale, group_scale_quant, super_group_zero, group_zero_quant) -> torch.Tensor:
    """pack_tensor for operands that are (or go) on the GPU; the packed bytes stay there."""
    q_type = T(int(q_type))
    if q_type in (T.Q3_K, T.Q6_K):
        return pack_on_device(q_type, qweight, super_group_scale, group_scale_quant)
    return pack_on_device(q_type, qweight, super_group_scale, group_scale_quant, super_group_zero, group_zero_quant)
