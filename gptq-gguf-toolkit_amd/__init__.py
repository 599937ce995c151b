"""gptq-gguf-toolkit_amd: MI355X-native GPTQ -> GGUF K-quant hot path.

Drop-in for the `quant/gptq` path of IST-DASLab/gptq-gguf-toolkit: same class protocol
(GPTQ.update/quantize/reset), same data.pth schema, same pack_Q*K byte layouts; every
numerical stage is a gfx950 HIP kernel behind the C ABI of include/gptq_gguf.h.
"""
from . import _cabi  # noqa: F401
from ._cabi import GQError, SO_PATH, build  # noqa: F401

__all__ = ["GQError", "SO_PATH", "build"]
