"""Minimal GGUF v3 writer (container only), written from the GGUF specification.

The reference delegates the container to gguf-py 0.17.1 (`GGUFWriter`, not vendored in its
tree and not installable here): header, key/value metadata, tensor infos, 32-byte aligned
tensor data.  PARITY UNPINNED for whole-file byte identity (no reference file exists to
compare against); the tensor PAYLOADS are the pinned part (packing_utils / gq_pack).
Layout (little endian):
  magic "GGUF" | u32 version=3 | u64 n_tensors | u64 n_kv
  n_kv x { string key | u32 value_type | value }
  n_tensors x { string name | u32 n_dims | u64 dims[n_dims] (ggml order: innermost first)
                | u32 ggml_type | u64 offset (from the start of the data section) }
  padding to `general.alignment` (32) | tensor data, each tensor padded to the alignment

Where the GGUF v3 specification leaves a choice, this writer makes the choice gguf-py's `GGUFWriter` makes (its published
behaviour, cited per rule as class.method; DESIGN.md section 0d lists them with the test that asserts each):
  R1  key/value pairs are written in INSERTION order and `general.architecture` is the first one (`GGUFWriter.__init__`
      calls `add_architecture()`; `write_kv_data_to_file` walks the dict);
  R2  a key added twice raises ValueError("Duplicated key name ...") (`add_key_value`); a tensor name added twice raises
      ValueError("Duplicated tensor name ...") (`add_tensor_info`);
  R3  the element type of an ARRAY is the type of its FIRST element by the Python type -- str -> STRING, bool -> BOOL,
      int -> INT32, float -> FLOAT32 (`GGUFValueType.get_type`) -- and every element must map to the same type
      (ValueError "All items in a GGUF array should be of the same type", `_pack_val`); an EMPTY array is not written at all
      (`add_array` returns early);
  R4  strings are UTF-8 with a u64 byte length and no terminator; BOOL is one byte 0 / 1 (`_pack_val`);
  R5  tensor infos AND tensor data are both in the order of the `add_tensor` calls (`write_ti_data_to_file` /
      `write_tensors_to_file` walk the same dict); a tensor's offset is relative to the start of the data section and
      advances by `ggml_pad(nbytes, alignment)`; dimensions are stored innermost first;
  R6  every pad byte is 0x00: between the tensor infos and the data section and after EVERY tensor including the last one
      (`write_padding` writes `bytes([0] * pad)`), so the file length is a multiple of the alignment;
  R7  the alignment is 32 and `general.alignment` is NOT written unless a custom alignment was asked for;
  R8  a uint8 array handed over with a `raw_dtype` is block bytes: its logical shape is `quant_shape_from_byte_shape`
      (`add_tensor`); fp16 / fp32 numpy arrays name their own type.
"""
import struct
from typing import Any, List, Sequence, Tuple

import numpy as np

GGUF_MAGIC = b"GGUF"
GGUF_VERSION = 3
ALIGNMENT = 32


class GGUFValueType:
    UINT8, INT8, UINT16, INT16, UINT32, INT32, FLOAT32, BOOL, STRING, ARRAY, UINT64, INT64, FLOAT64 = range(13)


class GGMLType:
    F32, F16 = 0, 1
    Q8_0 = 8
    Q2_K, Q3_K, Q4_K, Q5_K, Q6_K = 10, 11, 12, 13, 14
    BF16 = 30


# (block size in values, bytes per block)
GGML_QUANT_SIZES = {GGMLType.F32: (1, 4), GGMLType.F16: (1, 2), GGMLType.BF16: (1, 2), GGMLType.Q8_0: (32, 34),
                    GGMLType.Q2_K: (256, 84),
                    GGMLType.Q3_K: (256, 110), GGMLType.Q4_K: (256, 144), GGMLType.Q5_K: (256, 176),
                    GGMLType.Q6_K: (256, 210)}


def _s(b: str) -> bytes:
    e = b.encode("utf-8")
    return struct.pack("<Q", len(e)) + e


_SCALAR_FMT = {GGUFValueType.UINT8: "<B", GGUFValueType.INT8: "<b", GGUFValueType.UINT16: "<H",
               GGUFValueType.INT16: "<h", GGUFValueType.UINT32: "<I", GGUFValueType.INT32: "<i",
               GGUFValueType.FLOAT32: "<f", GGUFValueType.BOOL: "<?", GGUFValueType.UINT64: "<Q",
               GGUFValueType.INT64: "<q", GGUFValueType.FLOAT64: "<d"}


def value_type_of(v: Any) -> int:
    """gguf-py `GGUFValueType.get_type` (rule R3): the GGUF type a Python value is written as when none is given."""
    if isinstance(v, (str, bytes, bytearray)):
        return GGUFValueType.STRING
    if isinstance(v, (list, tuple)):
        return GGUFValueType.ARRAY
    if isinstance(v, float):
        return GGUFValueType.FLOAT32
    if isinstance(v, bool):  # before int: bool is an int in Python
        return GGUFValueType.BOOL
    if isinstance(v, int):
        return GGUFValueType.INT32
    raise ValueError(f"Unknown type: {type(v)}")


def _pack_value(vtype: int, v: Any, sub: int = None) -> bytes:
    if vtype == GGUFValueType.STRING:
        return _s(v)
    if vtype == GGUFValueType.ARRAY:
        out = struct.pack("<IQ", sub, len(v))
        return out + b"".join(_pack_value(sub, x) for x in v)
    return struct.pack(_SCALAR_FMT[vtype], v)


def quant_shape_from_byte_shape(shape: Sequence[int], ggml_type: int) -> Tuple[int, ...]:
    bs, ts = GGML_QUANT_SIZES[ggml_type]
    assert shape[-1] % ts == 0
    return (*shape[:-1], shape[-1] // ts * bs)


class QuantError(Exception):
    """gguf-py's `gguf.QuantError`: the tensor's row length is not a multiple of the type's block size."""


def quantize_q8_0(data: np.ndarray) -> np.ndarray:
    """`gguf.quants.quantize(data, Q8_0)` of gguf-py 0.17.1 (un-vendored; the reference calls it for the tensors GPTQ did
    not quantize when `--outtype q8_0`, pack_gptq_into_gguf.py:404-405,419), restated from its published algorithm,
    which is ggml-quants.c `quantize_row_q8_0_ref` in numpy: per block of 32 fp32 values `d = amax / 127` (fp32),
    `id = 1 / d` (0 when d == 0), `q = roundf(x * id)` -- ROUND HALF AWAY FROM ZERO, computed as
    `sign(x) * (floor(|v|) + floor(2 * (|v| - floor(|v|))))` -- stored as `block_q8_0 { fp16 d; int8 qs[32]; }`, 34 bytes.
    -> uint8 [..., n / 32 * 34].  PARITY: pinned against an independent scalar restatement of the C routine
    (tests/ggml_spec.py), not against gguf-py itself (not installable here)."""
    data = np.asarray(data)
    if data.shape[-1] % 32 != 0:
        raise QuantError(f"Can't quantize tensor with shape {data.shape} to Q8_0")
    blocks = np.ascontiguousarray(data, dtype=np.float32).reshape(-1, 32)
    d = np.abs(blocks).max(axis=1, keepdims=True) / np.float32(127)
    with np.errstate(divide="ignore"):
        inv = np.where(d == 0, np.float32(0), np.float32(1) / d).astype(np.float32)
    v = blocks * inv
    a = np.abs(v)
    fl = np.floor(a)
    qs = (np.sign(v) * (fl + np.floor(2 * (a - fl)))).astype(np.int8).view(np.uint8)
    out = np.concatenate([d.astype(np.float16).view(np.uint8), qs], axis=1)
    return out.reshape(*data.shape[:-1], data.shape[-1] // 32 * 34)


class GGUFWriter:
    def __init__(self, path: str, arch: str):
        self.path = path
        self.kv: List[Tuple[str, int, Any, int]] = []
        self.tensors: List[Tuple[str, Tuple[int, ...], int, np.ndarray]] = []
        self.add_string("general.architecture", arch)

    # ---- metadata
    def add(self, key, vtype, value, sub=None):
        if any(k == key for k, *_ in self.kv):  # gguf-py's GGUFWriter.add_key_value raises the same way
            raise ValueError(f"Duplicated key name {key!r}")
        self.kv.append((key, vtype, value, sub))

    def add_string(self, k, v): self.add(k, GGUFValueType.STRING, v)
    def add_uint32(self, k, v): self.add(k, GGUFValueType.UINT32, int(v))
    def add_float32(self, k, v): self.add(k, GGUFValueType.FLOAT32, float(v))
    def add_bool(self, k, v): self.add(k, GGUFValueType.BOOL, bool(v))

    def add_array(self, k, v, sub=None):
        """Rule R3.  `sub` given (this package always names it): it must be what gguf-py would infer from the elements --
        the reference calls `add_array(key, values)` / `add_token_types` / `add_token_scores` without a type."""
        v = list(v)
        if len(v) == 0:
            return  # gguf-py: `if len(val) == 0: return`
        inferred = value_type_of(v[0])
        if not all(value_type_of(x) == inferred for x in v[1:]):
            raise ValueError("All items in a GGUF array should be of the same type")
        if sub is not None and sub != inferred:
            raise ValueError(f"array {k!r}: element type {sub} asked for, gguf-py would write {inferred} for these values")
        self.add(k, GGUFValueType.ARRAY, v, inferred)

    # ---- tensors
    def add_tensor(self, name: str, data: np.ndarray, raw_dtype: int = None):
        """raw_dtype given: `data` is uint8 [.., nbytes] block bytes (reference pack_gptq_into_gguf.py:344-348)."""
        data = np.ascontiguousarray(data)
        if raw_dtype is None:
            raw_dtype = {np.dtype(np.float32): GGMLType.F32, np.dtype(np.float16): GGMLType.F16}[data.dtype]
            shape = data.shape
        else:
            shape = quant_shape_from_byte_shape(data.shape, raw_dtype) if data.dtype == np.uint8 else data.shape
        if any(n == name for n, *_ in self.tensors):  # rule R2
            raise ValueError(f"Duplicated tensor name {name!r}")
        self.tensors.append((name, tuple(int(x) for x in shape), int(raw_dtype), data))

    def add_tensor_lazy(self, name: str, shape, raw_dtype: int, producer):
        """A tensor whose bytes are made when the file is written (gguf-py has the same split: add_tensor_info now,
        write_tensor_data later).  `shape` is the LOGICAL shape; `producer()` returns a C-contiguous numpy array of exactly
        the type's byte count.  write() runs the producers on threads of their own (LAZY_WORKERS at a time, results in tensor
        order), a few tensors ahead of the file writes: a converter never holds more than `LAZY_DEPTH` payloads (pack_gptq_into_gguf.convert: ~5 GB otherwise)."""
        if any(n == name for n, *_ in self.tensors):  # rule R2
            raise ValueError(f"Duplicated tensor name {name!r}")
        shape = tuple(int(x) for x in shape)
        bs, ts = GGML_QUANT_SIZES[int(raw_dtype)]
        nbytes = int(np.prod(shape)) // bs * ts
        self.tensors.append((name, shape, int(raw_dtype), _Lazy(producer, nbytes)))

    LAZY_DEPTH = 4
    LAZY_WORKERS = 3

    def write(self, timing: dict = None):
        """`timing` (optional) receives seconds: "write" = file writes, "wait" = the writer waiting for a producer."""
        import queue
        import threading
        import time
        lazy = [d for *_, d in self.tensors if isinstance(d, _Lazy)]
        q: "queue.Queue" = queue.Queue(maxsize=self.LAZY_DEPTH)

        def produce():
            # LAZY_WORKERS producers run at a time (a payload's upload is a pageable copy bound by ONE host core: two or three
            # of them overlap), their results are handed over in tensor order, at most LAZY_DEPTH ahead of the file writes
            from concurrent.futures import ThreadPoolExecutor
            try:
                with ThreadPoolExecutor(max_workers=self.LAZY_WORKERS) as pool:
                    pending = []
                    for d in lazy:
                        pending.append(pool.submit(d.producer))
                        if len(pending) >= self.LAZY_WORKERS:
                            q.put(pending.pop(0).result())
                    for f in pending:
                        q.put(f.result())
            except BaseException as e:  # handed to the writing thread
                q.put(e)

        th = threading.Thread(target=produce, daemon=True) if lazy else None
        if th:
            th.start()
        t_write = t_wait = 0.0
        with open(self.path, "wb") as f:
            f.write(GGUF_MAGIC + struct.pack("<IQQ", GGUF_VERSION, len(self.tensors), len(self.kv)))
            for k, t, v, sub in self.kv:
                f.write(_s(k) + struct.pack("<I", t) + _pack_value(t, v, sub))
            off = 0
            for name, shape, gt, data in self.tensors:
                f.write(_s(name) + struct.pack("<I", len(shape)))
                f.write(b"".join(struct.pack("<Q", d) for d in reversed(shape)))  # ggml ne[0] = innermost
                f.write(struct.pack("<IQ", gt, off))
                off += (data.nbytes + ALIGNMENT - 1) // ALIGNMENT * ALIGNMENT
            pad = (-f.tell()) % ALIGNMENT
            f.write(b"\x00" * pad)
            for name, _, _, data in self.tensors:
                if isinstance(data, _Lazy):
                    t0 = time.perf_counter()
                    got = q.get()
                    t_wait += time.perf_counter() - t0
                    if isinstance(got, BaseException):
                        raise got
                    got = np.ascontiguousarray(got)
                    if got.nbytes != data.nbytes:
                        raise ValueError(f"tensor {name!r}: producer returned {got.nbytes} bytes, {data.nbytes} announced")
                    data = got
                t0 = time.perf_counter()
                f.write(data.reshape(-1).view(np.uint8).data)  # the array's own buffer: no tobytes() copy
                f.write(b"\x00" * ((-data.nbytes) % ALIGNMENT))
                t_write += time.perf_counter() - t0
        if th:
            th.join()
        if timing is not None:
            timing["write"] = timing.get("write", 0.0) + t_write
            timing["wait"] = timing.get("wait", 0.0) + t_wait


class _Lazy:
    def __init__(self, producer, nbytes):
        self.producer, self.nbytes = producer, nbytes


def parse_gguf(path: str):
    """-> (kv {key: (value, [value type ids])}, [(name, logical shape (outermost first), ggml type, absolute data
    offset, n_bytes)], file bytes).  Spec-level reader for this package's tests and the GGUF splitter."""
    buf = open(path, "rb").read()
    pos = 0

    def rd(fmt):
        nonlocal pos
        v = struct.unpack_from(fmt, buf, pos)
        pos += struct.calcsize(fmt)
        return v if len(v) > 1 else v[0]

    def rs():
        nonlocal pos
        n = rd("<Q")
        s = buf[pos:pos + n].decode("utf-8")
        pos += n
        return s

    def rv(t):
        if t == GGUFValueType.STRING:
            return rs(), [t]
        if t == GGUFValueType.ARRAY:
            sub, n = rd("<I"), rd("<Q")
            return [rv(sub)[0] for _ in range(n)], [t, sub]
        return rd(_SCALAR_FMT[t]), [t]

    assert buf[:4] == GGUF_MAGIC
    pos = 4
    ver, nt, nkv = rd("<I"), rd("<Q"), rd("<Q")
    assert ver == GGUF_VERSION
    kv = {}
    for _ in range(nkv):
        k = rs()
        kv[k] = rv(rd("<I"))
    infos = []
    for _ in range(nt):
        name = rs()
        nd = rd("<I")
        dims = [rd("<Q") for _ in range(nd)]
        gt, off = rd("<I"), rd("<Q")
        infos.append((name, tuple(reversed(dims)), gt, off))
    align = int(kv["general.alignment"][0]) if "general.alignment" in kv else ALIGNMENT
    data0 = (pos + align - 1) // align * align
    tensors = []
    for name, shape, gt, off in infos:
        bs, ts = GGML_QUANT_SIZES[gt]
        tensors.append((name, shape, gt, data0 + off, int(np.prod(shape)) // bs * ts))
    return kv, tensors, buf


def read_gguf(path: str):
    """Tiny reader (tests / verification): -> (kv dict, {name: (shape, ggml_type, raw bytes as np.uint8)})."""
    kv, tensors, buf = parse_gguf(path)
    return ({k: v for k, (v, _) in kv.items()},
            {name: (shape, gt, np.frombuffer(buf, np.uint8, n, off)) for name, shape, gt, off, n in tensors})
