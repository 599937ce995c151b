#!/usr/bin/env python3
"""GGUF splitter -- the EvoPress database producer of the reference (mapper/gguf_splitter.py:33-446): one directory per
GGUF tensor holding the tensor's raw bytes as "<bitwidth>[-<quantization>].pth" plus a metadata JSON, a manifest of
all tensors with the file's key/value metadata, and the layer database (name -> type / bit width / shape / offset).
Same file names, JSON keys and CLI (`model_path output_dir [--exact]`); the container is read by this package's
spec-level reader instead of gguf-py (not installable here), so only what this package writes (F32 / F16 / BF16 /
K-quants) is understood.  The HF-side split of the reference (`--hf-layers`: loads the model back through
transformers' GGUF loader) is not part of this file.  Pure file plumbing: no arithmetic of the hot path lives here.
"""
import argparse
import json
import time
from pathlib import Path
from typing import Dict, Union

import numpy as np

try:
    from .gguf_writer import GGML_QUANT_SIZES, parse_gguf
except ImportError:  # run as a script
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from gptq_gguf_toolkit_amd.gguf_writer import GGML_QUANT_SIZES, parse_gguf

TYPE_NAMES = {0: "F32", 1: "F16", 2: "Q4_0", 3: "Q4_1", 6: "Q5_0", 7: "Q5_1", 8: "Q8_0", 9: "Q8_1", 10: "Q2_K",
              11: "Q3_K", 12: "Q4_K", 13: "Q5_K", 14: "Q6_K", 15: "Q8_K", 30: "BF16"}  # gguf_splitter.py:42-50 (+BF16)
EXACT_BITS = {"F32": 32.0, "F16": 16.0, "BF16": 16.0, "Q4_0": 4.5, "Q4_1": 5.0, "Q5_0": 5.5, "Q5_1": 6.0, "Q8_0": 8.5,
              "Q8_1": 9.0, "Q2_K": 2.5625, "Q3_K": 3.4375, "Q4_K": 4.5, "Q5_K": 5.5, "Q6_K": 6.5625, "Q8_K": 8.5}  # :56-96


class GGUFSplitter:
    def __init__(self, model_path: str, output_dir: str, use_exact_bitwidth: bool = False):
        self.model_path, self.output_dir = Path(model_path), Path(output_dir)
        self.output_dir.mkdir(parents=True, exist_ok=True)
        self.use_exact_bitwidth = use_exact_bitwidth
        self.gguf_layer_database: Dict[str, dict] = {}

    def get_quantization_info(self, tensor_name: str, tensor_type: int) -> str:
        return TYPE_NAMES.get(tensor_type, f"UNKNOWN_{tensor_type}")

    def get_tensor_bit_width(self, quantization: str) -> float:
        return EXACT_BITS.get(quantization, 32.0)

    def extract_bitwidth_from_quantization(self, quantization: str) -> Union[int, float]:
        """gguf_splitter.py:98-122: the exact fractional width, or the integer class of the type."""
        if self.use_exact_bitwidth:
            return self.get_tensor_bit_width(quantization)
        for digit in "234568":
            if quantization.startswith(("Q" + digit, "IQ" + digit)):
                return int(digit)
        if quantization in ("F16", "F32"):
            return 16 if quantization == "F16" else 32
        return 1 if quantization.startswith("IQ1") else 4

    def _entries(self):
        kv, tensors, buf = parse_gguf(str(self.model_path))
        for name, shape, gt, off, nbytes in tensors:
            q = self.get_quantization_info(name, gt)
            yield name, shape, gt, off, nbytes, q, buf
        self._kv = kv

    def build_gguf_layer_database(self) -> Dict[str, dict]:
        db = {}
        for name, shape, gt, off, nbytes, q, _ in self._entries():
            db[name] = {"tensor_type": gt, "quantization": q, "bitwidth": self.extract_bitwidth_from_quantization(q),
                        "exact_bitwidth": self.get_tensor_bit_width(q), "shape": list(reversed(shape)),  # ggml ne order
                        "n_elements": int(np.prod(shape)), "n_bytes": nbytes, "data_offset": off}
        self.gguf_layer_database = db
        return db

    def split_gguf_model(self, overwrite_bitwidth=None):
        self.build_gguf_layer_database()
        manifest = {"model_info": {"original_file": self.model_path.name, "total_tensors": len(self.gguf_layer_database),
                                   "split_timestamp": None, "use_exact_bitwidth": self.use_exact_bitwidth},
                    "metadata": {}, "layers": {}}
        n = 0
        for name, shape, gt, off, nbytes, q, buf in self._entries():
            n += 1
            bitwidth = self.extract_bitwidth_from_quantization(q)
            prefix = f"{bitwidth}" if isinstance(bitwidth, float) and bitwidth != int(bitwidth) else f"{int(bitwidth)}"
            if self.use_exact_bitwidth:
                prefix = f"{prefix}-{q}"
            layer_dir = self.output_dir / name
            layer_dir.mkdir(parents=True, exist_ok=True)
            (layer_dir / f"{prefix}.pth").write_bytes(buf[off:off + nbytes])  # raw bytes, not a torch pickle (:378-380)
            bs, ts = GGML_QUANT_SIZES[gt]
            if bs > 1:
                np_dtype, np_shape = "uint8", [*shape[:-1], shape[-1] // bs * ts]
            else:
                np_dtype, np_shape = {0: "float32", 1: "float16", 30: "uint16"}[gt], list(shape)
            common = {"type": gt, "quantization": q, "bitwidth": bitwidth, "exact_bitwidth": self.get_tensor_bit_width(q),
                      "shape": list(reversed(shape)), "n_elements": int(np.prod(shape))}
            (layer_dir / f"{prefix}-metadata.json").write_text(json.dumps({"tensor_info": {
                "name": name, **common, "n_bytes": nbytes, "data_offset_original": off, "data_filename": f"{prefix}.pth",
                "np_dtype": np_dtype, "np_shape": np_shape}}, indent=2))
            layer = manifest["layers"].setdefault(name, {"original_name": name, "dims": list(reversed(shape)), "bitwidths": {}})
            layer["bitwidths"][str(bitwidth)] = {"filename": f"{prefix}.pth", "metadata_filename": f"{prefix}-metadata.json",
                                                 **common, "size_bytes": nbytes, "data_offset": off}
        for key, (value, types) in self._kv.items():
            manifest["metadata"][key] = {"types": types, "value": value}
        manifest["model_info"]["split_timestamp"] = time.time()
        manifest["model_info"]["processed_tensors"] = n
        (self.output_dir / "manifest.json").write_text(json.dumps(manifest, indent=2))
        (self.output_dir / "gguf_layer_database.json").write_text(json.dumps(self.gguf_layer_database, indent=2))
        return manifest


def main(argv=None):
    p = argparse.ArgumentParser(description="Split a GGUF model into per-tensor directories (EvoPress database)")
    p.add_argument("model_path", help="Path to input GGUF model")
    p.add_argument("output_dir", help="Directory to store split layers")
    p.add_argument("--exact", action="store_true", help='exact fractional bit widths in the file names ("4.5-Q4_K.pth")')
    a = p.parse_args(argv)
    m = GGUFSplitter(a.model_path, a.output_dir, use_exact_bitwidth=a.exact).split_gguf_model()
    print(f"GGUF split complete! {m['model_info']['processed_tensors']} tensors into {len(m['layers'])} layer directories")


if __name__ == "__main__":
    main()
