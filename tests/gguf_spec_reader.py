"""An independent, minimal GGUF v3 key/value reader written from the GGUF specification (ggml/docs/gguf.md), used ONLY by
the tests to read back what gptq_gguf_toolkit_amd.gguf_writer wrote (the package's own read_gguf is not trusted to check
itself).  header := u32 magic "GGUF", u32 version, u64 n_tensors, u64 n_kv; kv := string key, u32 type, value;
string := u64 length + utf-8 bytes; array := u32 element type, u64 count, elements."""
import struct

_SCALAR = {0: "<B", 1: "<b", 2: "<H", 3: "<h", 4: "<I", 5: "<i", 6: "<f", 7: "<?", 10: "<Q", 11: "<q", 12: "<d"}


def read_kv(path):
    buf = open(path, "rb").read()
    assert buf[:4] == b"GGUF"
    version, n_tensors, n_kv = struct.unpack_from("<IQQ", buf, 4)
    assert version == 3
    pos = 24

    def string():
        nonlocal pos
        (n,) = struct.unpack_from("<Q", buf, pos)
        pos += 8
        s = buf[pos:pos + n].decode("utf-8")
        pos += n
        return s

    def value(t):
        nonlocal pos
        if t == 8:
            return string()
        if t == 9:
            et, cnt = struct.unpack_from("<IQ", buf, pos)
            pos += 12
            return [value(et) for _ in range(cnt)]
        fmt = _SCALAR[t]
        (v,) = struct.unpack_from(fmt, buf, pos)
        pos += struct.calcsize(fmt)
        return v

    kv, types = {}, {}
    for _ in range(n_kv):
        k = string()
        (t,) = struct.unpack_from("<I", buf, pos)
        pos += 4
        if t == 9:
            types[k] = ("array", struct.unpack_from("<I", buf, pos)[0])
        else:
            types[k] = t
        kv[k] = value(t)
    return kv, types, n_tensors
