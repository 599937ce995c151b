"""An independent, minimal GGUF v3 serialiser written from the GGUF specification (ggml/docs/gguf.md), used ONLY to
pin the bytes gptq_gguf_toolkit_amd.gguf_writer produces (the reference delegates the container to gguf-py 0.17.1,
which is neither vendored nor installable here).  Different construction on purpose: one flat bytearray, explicit
little-endian encoders per value type, offsets computed in a separate pass.

  file   := header kv* tensor_info* pad(ALIGN) tensor_data*
  header := u32 magic(0x46554747 "GGUF") u32 version(3) u64 n_tensors u64 n_kv
  kv     := string key, u32 value_type, value
  string := u64 length, utf-8 bytes (no terminator)
  array  := u32 element_type, u64 count, elements
  info   := string name, u32 n_dims, u64 ne[n_dims] (ne[0] = innermost), u32 ggml_type, u64 offset (from data start)
  data   := every tensor padded with zeros to ALIGN = 32 (general.alignment default)
"""
ALIGN = 32
VT = {"u8": 0, "i8": 1, "u16": 2, "i16": 3, "u32": 4, "i32": 5, "f32": 6, "bool": 7, "str": 8, "arr": 9, "u64": 10,
      "i64": 11, "f64": 12}
GGML = {"F32": 0, "F16": 1, "Q2_K": 10, "Q3_K": 11, "Q4_K": 12, "Q5_K": 13, "Q6_K": 14, "BF16": 30}
BLOCK = {"F32": (1, 4), "F16": (1, 2), "BF16": (1, 2), "Q2_K": (256, 84), "Q3_K": (256, 110), "Q4_K": (256, 144),
         "Q5_K": (256, 176), "Q6_K": (256, 210)}


def _le(n, width, signed=False):
    return int(n).to_bytes(width, "little", signed=signed)


def _string(s):
    b = s.encode("utf-8")
    return _le(len(b), 8) + b


def _scalar(kind, v):
    import struct
    if kind == "f32":
        return struct.pack("<f", v)
    if kind == "f64":
        return struct.pack("<d", v)
    if kind == "bool":
        return b"\x01" if v else b"\x00"
    if kind == "str":
        return _string(v)
    width = {"u8": 1, "i8": 1, "u16": 2, "i16": 2, "u32": 4, "i32": 4, "u64": 8, "i64": 8}[kind]
    return _le(v, width, signed=kind[0] == "i")


def serialise(kvs, tensors):
    """kvs: [(key, kind, value)] with kind in VT (arrays: kind = ("arr", element_kind));
    tensors: [(name, logical_shape (outermost first), ggml type name, raw bytes)]."""
    out = bytearray()
    out += _le(0x46554747, 4) + _le(3, 4) + _le(len(tensors), 8) + _le(len(kvs), 8)
    for key, kind, value in kvs:
        out += _string(key)
        if isinstance(kind, tuple):
            out += _le(VT["arr"], 4) + _le(VT[kind[1]], 4) + _le(len(value), 8)
            for v in value:
                out += _scalar(kind[1], v)
        else:
            out += _le(VT[kind], 4) + _scalar(kind, value)
    offset = 0
    for name, shape, tname, raw in tensors:
        per, nbytes = BLOCK[tname]
        n = 1
        for d in shape:
            n *= d
        assert n % per == 0 and len(raw) == n // per * nbytes, (name, shape, len(raw))
        out += _string(name) + _le(len(shape), 4)
        for d in reversed(shape):
            out += _le(d, 8)
        out += _le(GGML[tname], 4) + _le(offset, 8)
        offset += -(-len(raw) // ALIGN) * ALIGN
    out += b"\x00" * (-len(out) % ALIGN)
    for _, _, _, raw in tensors:
        out += bytes(raw) + b"\x00" * (-len(raw) % ALIGN)
    return bytes(out)
